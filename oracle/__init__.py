"""CPU oracle for the PINN-elastodynamics hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``pinn_elastodynamics_amd/`` may import
this package: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` are allowed to, and there only as the
checker / the reported CPU baseline, never as the thing shipped.

Pinning status (see DESIGN.md, section "Oracle"):
  * The reference (TF 1.10 + pyDOE scripts under /root/reference) cannot be
    imported or built in this image, and it holds no tests / golden outputs.
  * The restatement is therefore pinned by (1) the reference's committed weight
    pickles as exact inputs, (2) the known answer "PDE residual ~ 0 at trained
    weights" (any sign / coefficient / derivative-pairing error makes it O(1)),
    (3) the reference's FEM frames (physics sanity bands) and (4) two
    independently written differentiation routes (closed-form forward-tangent
    numpy vs. reverse-mode torch autograd written op-for-op like the TF1 graph)
    that agree to ~1e-13.
  * What stays unpinned: TF1's RNG stream (initial weights), pyDOE's LHS point
    sets, Adam/L-BFGS trajectories.  Parity is defined on identical weights and
    identical point sets supplied as inputs.
"""
