"""oracle/ — CPU restatement of the reference's sampling hot path.  TEST INFRASTRUCTURE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
anything from this package, and only as the checker / the reported CPU baseline — never as
the thing shipped.  The product (`esmdiff_amd`) never imports it and fails loudly when
libesmdiff_hip.so is missing.

Parts
  sampler_ref.py   torch-CPU float32 restatement, op for op, of the reference's sampler
                   (model.py / noise_utils.py / net.py::TimestepEmbedder / sample_esmdiff.py batching /
                   eval_utils.merge_pdbfiles).  PINNED: reproduces tests/golden/g1..g8, which were
                   produced by importing the reference's own code (tests/golden/make_goldens.py).
  csrc/sampler_oracle.c
                   plain-C restatement of one update in the canonical float-op order the HIP kernel
                   uses (bit-exact gate).  PINNED through g3..g6 (ids exact, log-probs <= 4 ulp).
  esm3_ref.py      torch-CPU float32 restatement of the ESM3-open network as CustomizedESM3 wires it
                   (net.py:322-483).  PARITY UNPINNED: the arithmetic lives in the third-party package
                   esm==3.0.4 (requirements.txt:30) which is neither vendored in the reference nor
                   installable here; it is restated from the published architecture (SURVEY.md
                   Appendix A) and anchored only on the reference's call sites and in-tree
                   hyper-parameters (net.py:325-346, mdlm.yaml:26-58).
  gibbs_ref.py     torch restatement of the per-prompt half of esm's iterative_sampling_raw (SURVEY.md
                   Appendix B).  PARITY UNPINNED for the same reason.
  geom_ref.py      torch restatement of esm's build_affine3d_from_coordinates and of block 0's
                   GeometricReasoningOriginalImpl (call sites net.py:433-441, :468).
                   PARITY UNPINNED; checked for what must hold regardless (rigid-motion invariance, exact
                   zero branch without coordinates, frameless residues inert: tests/test_geom_cpu.py).
  metrics_ref.py   numpy restatement of the ensemble metrics of eval_utils.py (js_pwd, js_rg, validity,
                   bonding_validity, with numpy.histogram / scipy jensenshannon written out).  PINNED: reproduces
                   tests/golden/g9_metrics.npz (made by the reference's own functions) to 1e-16.
  encoder_ref.py   torch restatement of esm's StructureTokenEncoder (kNN neighbourhoods -> geometric blocks ->
                   codebook; call site models/utils.py:136-137).  PARITY UNPINNED; rigid-motion invariance checked.
  decoder_ref.py   torch restatement of esm's StructureTokenDecoder backbone path (embed -> block stack ->
                   Dim6RotStructureHead; call site sample_esmdiff.py:40-61).  PARITY UNPINNED.
"""
