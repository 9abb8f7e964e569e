"""CPU oracle for the uncertainty-rendering hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may.  The product path (``uncertainty-nerf-gs_amd``) fails loudly when the HIP
library is missing instead of falling back to anything in here.

Parity status ("Oracle pinning"):
  * reference-pinned  : create_mlp topology, ause, auce, sample_laplace,
                        ComputeWeightsModule (= get_weights), ensemble
                        aggregation -- checked against the *imported* reference
                        (tests/golden/make_golden.py) and frozen as fixtures.
  * parity unpinned   : everything that restates nerfstudio 1.1.0 / gsplat
                        0.1.11 behaviour (hash grid, samplers, renderers,
                        projection, rasteriser).  Those packages are not under
                        /root/reference and not installed; the restatement
                        follows their published algorithms and the reference's
                        call sites.
"""
