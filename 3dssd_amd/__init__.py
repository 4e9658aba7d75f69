"""3dssd_amd: MI355X (gfx950) implementation of the 3DSSD set-abstraction hot path behind the
reference's operator API.  The directory name is not a Python identifier; import it with
importlib.import_module("3dssd_amd"), or put this directory on sys.path and use the reference's own
import lines (`from utils.tf_ops.sampling.tf_sampling import *`)."""
