"""Placeholder: run_univl_amd.py must have replaced this module in sys.modules before the training script imports it."""
raise ImportError("stand-in modules.%s imported: the launcher shim did not install univl_amd" % __name__.split(".")[-1])
