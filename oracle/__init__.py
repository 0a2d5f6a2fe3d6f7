"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the Touch-GS hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it, and only as the checker.

PARITY UNPINNED for the rasterizer rows: the reference tree
(/root/reference) does not contain the rasterizer or the training model -- they live in the
un-vendored, empty submodule ``nerfstudio/`` (reference ``.gitmodules:7-9``) and its
third-party CUDA dependency; the only call sites are ``scripts/train_bunny_real.sh:52`` etc.
The oracle is therefore a restatement of the *published* 3DGS / EWA-splatting algorithm as
fixed in SURVEY.md Appendix B.  Two independent restatements are kept and cross-checked:
``torch_oracle.py`` (vectorised, differentiable, fp64) and ``ref_raster.c`` (scalar C).

PINNED rows: the tactile/vision depth fusion (a14) and touch back-projection follow
``utils/fuse_touch_vision.py`` / ``utils/create_point_cloud_from_touches.py`` and are pinned by
golden vectors generated from the reference itself (``tests/golden/make_fusion_golden.py``).
"""
