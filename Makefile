# (-Wl,-Bsymbolic: the libraries share inline functions of rt_prep.h; each must call its own copy, not whichever
# library the process happened to load first.)
# Builds the shipped libraries in-tree (rt_amd/lib, rt_amd/bin) and, separately, the test-only checkers.
#   make            -> librt_hip.so (hipcc, gfx950), librt_host.so + rt_cli (g++)
#   make checkers   -> oracle/librt_oracle.so and tests/emu/libcore_emu.so (never linked into the above)
HIPCC    ?= hipcc
CXX      ?= g++
ARCH     ?= gfx950
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt -Iinclude -Irt_amd/csrc
CXXFLAGS := -O2 -std=c++17 -ffp-contract=off -fPIC -Wall -Wextra -Iinclude -Irt_amd/host

LIB := rt_amd/lib
BIN := rt_amd/bin
CORE_HDRS := rt_amd/csrc/rt_core.h rt_amd/csrc/rt_prep.h include/rt_hip.h
HOST_HDRS := $(wildcard rt_amd/host/*/*.hpp) include/rt_hip.h include/rt_host.h

all: $(LIB)/librt_hip.so $(LIB)/librt_host.so $(BIN)/rt_cli

$(LIB)/librt_hip.so: rt_amd/csrc/rt_hip.hip $(CORE_HDRS)
	@mkdir -p $(LIB)
	$(HIPCC) $(HIPFLAGS) -shared -Wl,-Bsymbolic rt_amd/csrc/rt_hip.hip -o $@

$(LIB)/librt_host.so: rt_amd/host/rt_host_capi.cpp rt_amd/host/rt_hip/demo_frame_hip.cpp $(HOST_HDRS) $(LIB)/librt_hip.so
	$(CXX) $(CXXFLAGS) -shared rt_amd/host/rt_host_capi.cpp rt_amd/host/rt_hip/demo_frame_hip.cpp \
	    -L$(LIB) -lrt_hip -Wl,-rpath,'$$ORIGIN' -o $@

$(BIN)/rt_cli: rt_amd/host/cli/main.cpp $(HOST_HDRS) $(LIB)/librt_host.so
	@mkdir -p $(BIN)
	$(CXX) $(CXXFLAGS) rt_amd/host/cli/main.cpp -L$(LIB) -lrt_host -lrt_hip \
	    -Wl,-rpath,'$$ORIGIN/../lib' -pthread -o $@

# Profiling build of the kernels with region probes (tools/region_profile.py); never shipped.
prof: rt_amd/lib_prof/librt_hip.so rt_amd/lib_prof/librt_host.so
rt_amd/lib_prof/librt_host.so: $(LIB)/librt_host.so
	@mkdir -p rt_amd/lib_prof
	cp $(LIB)/librt_host.so $@
rt_amd/lib_prof/librt_hip.so: rt_amd/csrc/rt_hip.hip $(CORE_HDRS)
	@mkdir -p rt_amd/lib_prof
	$(HIPCC) $(HIPFLAGS) -DRT_PROFILE_REGIONS -shared -Wl,-Bsymbolic rt_amd/csrc/rt_hip.hip -o $@

checkers: oracle/librt_oracle.so oracle/librt_oracle_o3.so oracle/liboracle_frame.so tests/emu/libcore_emu.so

oracle/librt_oracle.so: oracle/rt_oracle.c oracle/rt_oracle.h
	$(MAKE) -C oracle librt_oracle.so

oracle/librt_oracle_o3.so: oracle/rt_oracle.c oracle/rt_oracle.h
	$(MAKE) -C oracle librt_oracle_o3.so

oracle/liboracle_frame.so: oracle/frame_cpu.cpp oracle/librt_oracle.so $(HOST_HDRS)
	$(MAKE) -C oracle liboracle_frame.so

tests/emu/libcore_emu.so: tests/emu/core_emu.cpp $(CORE_HDRS)
	$(CXX) -O2 -std=c++17 -ffp-contract=off -fPIC -shared -Wl,-Bsymbolic -Iinclude -Irt_amd/csrc tests/emu/core_emu.cpp -o $@

# Sanitizer builds of everything that runs on the host (the checkers, the host mirror and its C ABI), and the CPU
# tests under them.  For the build container only: no GPU is involved (librt_host.so links the ordinary librt_hip.so,
# which loads without one).  Python itself is not instrumented, so the sanitizer runtime is preloaded.
#   make san        AddressSanitizer + UndefinedBehaviorSanitizer, then ThreadSanitizer
SAN_TESTS := tests/test_oracle_golden.py tests/test_oracle_properties.py tests/test_emu_vs_oracle.py tests/test_host.py tests/test_sharding_runs.py tests/test_launch_plan.py
SAN_HOST_SRC := rt_amd/host/rt_host_capi.cpp rt_amd/host/rt_hip/demo_frame_hip.cpp
san: san-asan san-tsan

build/san-%/.built: oracle/rt_oracle.c oracle/rt_oracle.h oracle/frame_cpu.cpp tests/emu/core_emu.cpp $(SAN_HOST_SRC) $(CORE_HDRS) $(HOST_HDRS) $(LIB)/librt_hip.so
	@mkdir -p $(@D)
	gcc -O1 -g -std=c99 -ffp-contract=off -fno-fast-math -fPIC -pthread $(SAN_FLAGS) -shared oracle/rt_oracle.c -o $(@D)/librt_oracle.so -lm
	cp $(@D)/librt_oracle.so $(@D)/librt_oracle_o3.so
	$(CXX) -O1 -g -std=c++17 -ffp-contract=off -fPIC -pthread $(SAN_FLAGS) -Ioracle -Iinclude -Irt_amd/host -shared oracle/frame_cpu.cpp \
	    -L$(@D) -lrt_oracle -Wl,-rpath,'$$ORIGIN' -o $(@D)/liboracle_frame.so
	$(CXX) -O1 -g -std=c++17 -ffp-contract=off -fPIC $(SAN_FLAGS) -shared -Wl,-Bsymbolic -Iinclude -Irt_amd/csrc tests/emu/core_emu.cpp -o $(@D)/libcore_emu.so
	cp $(LIB)/librt_hip.so $(@D)/librt_hip.so
	$(CXX) -O1 -g -std=c++17 -ffp-contract=off -fPIC $(SAN_FLAGS) -Iinclude -Irt_amd/host -shared $(SAN_HOST_SRC) \
	    -L$(@D) -lrt_hip -Wl,-rpath,'$$ORIGIN' -o $(@D)/librt_host.so
	@touch $@

build/san-asan/.built: SAN_FLAGS := -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer
build/san-tsan/.built: SAN_FLAGS := -fsanitize=thread -fno-omit-frame-pointer

san-asan: build/san-asan/.built
	LD_PRELOAD=$$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
	    RT_CHECKER_LIB_DIR=$(CURDIR)/build/san-asan RT_AMD_LIB_DIR=$(CURDIR)/build/san-asan python -m pytest $(SAN_TESTS) -x -q -p no:cacheprovider

san-tsan: build/san-tsan/.built
	LD_PRELOAD=$$(gcc -print-file-name=libtsan.so) TSAN_OPTIONS=halt_on_error=1:report_signal_unsafe=0 \
	    RT_CHECKER_LIB_DIR=$(CURDIR)/build/san-tsan RT_AMD_LIB_DIR=$(CURDIR)/build/san-tsan python -m pytest $(SAN_TESTS) -x -q -p no:cacheprovider

clean:
	rm -f $(LIB)/*.so $(BIN)/rt_cli tests/emu/libcore_emu.so
	rm -rf build
	$(MAKE) -C oracle clean

.PHONY: all checkers prof clean san san-asan san-tsan
