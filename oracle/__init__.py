"""CPU oracle for the GANgealing hot path.  TEST INFRASTRUCTURE ONLY.

Everything in this package is a CPU restatement of the reference algorithm
(wpeebles/gangealing) used as a *checker*.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The product path (``gangealing_amd``) never does: it is HIP-only
and fails loudly when the gfx950 library is missing.

Pinning status: the reference ships no golden vectors or tests,
so the oracle is pinned against outputs of the reference's own pure-PyTorch CPU
bodies imported from ``/root/reference`` in the authoring container
(``oracle/make_golden.py`` -> ``tests/golden/*.npz``, checked by
``tests/test_oracle_golden.py``).  The arithmetic that lives in un-vendored
PyTorch ATen (grid_sample / affine_grid / interpolate / conv2d) is pinned
against this exact torch build (2.10.0+rocm7.0 CPU kernels); the reference
itself only pins ``pytorch>=1.10.1`` (environment.yml:10).
"""
