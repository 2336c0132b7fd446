from cdcmdr_amd.model.autoint import *  # noqa: F401,F403
from cdcmdr_amd.model import autoint as _m

globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
