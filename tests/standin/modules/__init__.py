"""Stand-in for the reference's `modules` package, used only by tests/test_shim_gpu.py on the GPU box (where the reference
checkout does not exist).  It reproduces the import-time friction of the real package -- nothing else."""
