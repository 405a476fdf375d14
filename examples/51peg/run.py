#!/usr/bin/env python3
"""The reference's 51 Peg example (evidence/examples/51Peg/run.py) on the MI355X engine: read the
config, build the model, hand the callbacks to a sampler.  Uses UltraNest (vectorized) when it is
installed, otherwise the in-repo batched driver with the proposal walk on the GPU.  Needs a GPU."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from evidence_amd import GpuRVModel                                   # noqa: E402
from evidence_amd.callbacks import make_ultranest_callbacks, wrapped_params   # noqa: E402
from evidence_amd.config import read_config                           # noqa: E402

nplanets = 1
rundict, datadict, priordict, fixedpardict = read_config(Path(__file__).with_name("config_51peg.py"), nplanets)
model = GpuRVModel(fixedpardict, datadict, list(priordict), priordict=priordict)
prior, loglike = make_ultranest_callbacks(model, vectorized=True)

try:
    from ultranest import ReactiveNestedSampler
    sampler = ReactiveNestedSampler(model.parnames, loglike, prior, vectorized=True,
                                    wrapped_params=wrapped_params(model.parnames))
    sampler.run(min_num_live_points=400)
    sampler.print_results()
except ImportError:
    from evidence_amd.nested import run_nested_slice
    res = run_nested_slice(prior, loglike, model.ndim, nlive=400, dlogz=0.5, max_calls=20_000_000,
                           wrapped=wrapped_params(model.parnames), seed=int(sys.argv[1]) if len(sys.argv) > 1 else 0,
                           prior_loglike=model.prior_loglike_batch, walker=model.slice_walk)
    print(f"{rundict['target']}: ln Z = {res.logz:.3f} +- {res.logzerr:.3f} "
          f"({res.niter} iterations, {res.ncall} likelihood calls)")
