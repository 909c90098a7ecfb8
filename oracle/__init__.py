"""CPU oracle for the UniMP / open_flamingo training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``unimp_amd/`` may import this package;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and there only as the checker / the timed CPU baseline.

It is a plain PyTorch fp32 (CPU) restatement of the arithmetic that the
reference executes for one optimizer step:

* ``oracle.vit``        -- CLIP ViT vision tower
                            (UniMP/xformers_model/clip.py:50-206,416-481 for the math;
                             open_clip parameter naming per SURVEY.md Appendix A.4)
* ``oracle.lm``         -- GPT-NeoX / OPT causal-LM towers
                            (transformers gpt_neox / opt modelling files; call sites
                             UniMP/mmrec.py:475-524)
* ``oracle.mpt``        -- MPT tower of OpenFlamingo-9B (ALiBi, bias-free, tied head; transformers' mpt modelling
                            file; call site UniMP/mmrec.py:515-524); cross-checked like ``oracle.lm``
                            (tests/golden/mpt_tiny_h{4,6}.npz)
* ``oracle.llama``      -- RMSNorm / RoPE / SwiGLU / causal attention of the in-tree
                            UniMP/xformers_model/llama.py:101-308
* ``oracle.flamingo``   -- Flamingo, PerceiverResampler, MaskedCrossAttention,
                            GatedCrossAttentionBlock, FlamingoLayer
                            (pip open-flamingo==2.0.1, requirements.txt:36; source is NOT in
                             /root/reference -- restated from the published algorithm,
                             SURVEY.md Appendix A; call sites UniMP/mmrec.py:20-22,476-524,177-181)
* ``oracle.train_step`` -- label mask, weighted focal CE, AdamW grouping, clip, schedule
                            (UniMP/mmrec.py:143-168,190-213,247-256,609-631,676-697)

* ``oracle.preprocess`` -- image half of the input pipeline: Pillow's 8-bit bicubic resampler (numpy restatement),
                            ToTensor, Normalize (UniMP/pipeline/mm_utils/rec_dataset.py:30-31,91-107;
                             transforms.py:102-136)

Parity pinning status (see DESIGN.md "Oracle"):
  - preprocess:   PINNED against Pillow itself (the third-party library the reference's transform calls,
                  requirements.txt:17; 12.2.0 here): tests/golden/preprocess_pillow.npz + live comparison,
                  generator oracle/make_golden_preprocess.py.  collate_fn is checked against the reference's own
                  collate_rec.py output (tests/golden/collate_rec.npz, same generator); the rec-task dataset reader
                  against the reference's own RecDataset class (tests/golden/rec_dataset.npz).
  - train_step:   PINNED against the reference itself (UniMP/mmrec.py:train_one_epoch run
                  in the build container with stubbed third-party imports; fixtures in
                  tests/golden/train_step_*.npz, generator oracle/make_golden.py).
  - vit / llama:  PINNED against the in-tree UniMP/xformers_model/{clip,llama}.py
                  (fixtures tests/golden/clip_tiny.npz, llama_tiny.npz).
  - lm:           cross-checked against the installed transformers GPT-NeoX / OPT
                  (third-party, same arithmetic as the reference's pinned 4.29; fixtures
                  tests/golden/neox_tiny.npz, opt_tiny.npz).
  - flamingo:     PARITY UNPINNED -- open-flamingo 2.0.1 is absent from /root/reference and
                  from this image and the reference holds no test or golden vector at that
                  boundary.  Anchored only by architecture known-answer tests
                  (tests/test_oracle_kat.py).
"""
