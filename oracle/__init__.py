"""CPU checker (test infrastructure).  Only tests, the smoke check and the benchmark's cpu_baseline leg import this."""
