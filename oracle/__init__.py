"""TEST INFRASTRUCTURE ONLY - never imported by the product package.

`oracle/` holds the CPU checkers for the MI355X ViLBERT hot path:

* ``ref_loader``      - imports the *real* reference model from /root/reference (only in the
                        build container; the path does not exist on the GPU box).
* ``vilbert_oracle``  - a from-scratch functional fp32/fp64 CPU restatement of the reference
                        forward (vilbert/vilbert.py), pinned against the real reference by
                        tests/test_oracle_vs_reference.py and by the fixtures in tests/golden/.
* ``synth``           - seeded weights / synthetic inputs shared by oracle, fixtures and bench.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
