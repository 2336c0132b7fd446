from cdcmdr_amd.model.dcn import *  # noqa: F401,F403
from cdcmdr_amd.model import dcn as _m

globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
