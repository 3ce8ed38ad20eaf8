"""A stand-in for the engine behind bench.py's control flow (tests/test_bench_control_flow.py): same Python surface as
ema_amd.engine / ema_amd.stream, candidates from the CPU oracle, no GPU.  TEST INFRASTRUCTURE -- it exists so that bench.py's
multi-rank paths (collectives, exits) run under gloo before a real node sees them; it is never a measured or shipped path."""
