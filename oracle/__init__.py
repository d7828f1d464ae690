"""oracle/ — CPU restatement of the reference's algorithm for the DxMI hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under diffusion-by-maxentirl_amd/ (the product) imports this
package; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as the
checker / the timed CPU baseline.

Form: torch-CPU fp32 (floating-point network math) + numpy float64 (schedule construction) +
integer index arithmetic, written from the reference's behaviour, each function citing the
reference file:line it follows (paths relative to the reference root).  The reference is pure
Python, so there is no C source to compile into oracle/_ref; instead the oracle is PINNED by golden
vectors generated from the reference itself in the build container
(tests/golden/make_golden.py -> tests/golden/*.npz; see DESIGN.md "Oracle").

Every network function takes `prec`, a Precision object: Precision("fp32") is the pinned
reference arithmetic; Precision("bf16") rounds operands/activations to bf16 at exactly the points
where the HIP pipeline stores bf16, which lets the GPU parity tests use tight tolerances.
"""
from .precision import Precision  # noqa: F401
