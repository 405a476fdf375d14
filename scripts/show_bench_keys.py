#!/usr/bin/env python3
"""Print the figures of a bench.py JSON line that the prior-stage / walk experiments look at."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("value", "prior_plus_loglike_evals_per_s", "prior_plus_loglike_one_launch_evals_per_s",
          "small_batch_prior_plus_loglike", "nested_sampling_end_to_end", "host_roundtrip_evals_per_s"):
    print(k, d.get(k))
