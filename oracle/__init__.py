"""CPU oracle for the PAAC hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This package is a CPU restatement (numpy, float64/float32) of the algorithm the
reference implements in paac.py / actor_learner.py / networks.py /
policy_v_network.py / environment.py / atari_emulator.py.  Each function cites
the reference file:line it follows.

Rules (enforced by tests/test_cabi_and_hostlogic.py::test_product_never_imports_oracle):
  * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
    import anything from here;
  * nothing under paac_amd/ imports it: the product path is HIP-only and fails
    loudly when libpaac_hip.so is missing.

Pinning status (see DESIGN.md "Oracle"):
  * rollout order, auto-reset, reward clip, masks, float64 n-step return scan,
    t-major flattening, lr schedule, global_step accounting, sampler RNG
    consumption, FramePool/ObservationPool: PINNED by tests/golden/*.npz, which
    were captured by running the reference's own train() loop
    (tests/golden/make_golden.py).
  * forward / loss / gradients / clip / RMSProp: the arithmetic lives in
    TensorFlow 1.0.1, which is absent here and has no golden vectors in the
    reference -> "parity unpinned" at that boundary; the restatement follows
    the .py sources + the constants frozen in pretrained/*/checkpoints/*.meta
    and is cross-checked against torch float64 autograd.
  * nearest resize: PIL NEAREST (what scipy.misc.imresize(interp='nearest')
    called); LUTs regenerated from PIL in tests wherever PIL is importable.
"""
