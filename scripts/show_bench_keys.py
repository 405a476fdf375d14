#!/usr/bin/env python3
"""Print the figures of bench.py JSON lines that the kernel / prior-stage / walk experiments look at (one file per argument)."""
import json, sys
for path in sys.argv[1:]:
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
    except (OSError, ValueError, IndexError) as exc:
        print(path, "unreadable:", exc)
        continue
    r = d.get("roofline", {})
    print(f"{path}: value {d.get('value'):.4g}  ms/step {d.get('ms_per_step')}  kernel_ms_timed_region {r.get('kernel_ms_timed_region')}  "
          f"valu_frac {r.get('frac')}  prewarm {d.get('config', {}).get('prewarm')}")
    for k in ("prior_plus_loglike_evals_per_s", "prior_plus_loglike_one_launch_evals_per_s", "loglike_alone_on_those_points_evals_per_s",
              "two_lane_pipelined_evals_per_s", "small_batch_prior_plus_loglike", "nested_sampling_end_to_end",
              "host_roundtrip_evals_per_s", "host_cube_to_logl_262144_rows", "sharded_configs_one_of_8_shards", "extras_failed"):
        if k in d:
            print("   ", k, d[k])
