from cdcmdr_amd.model.dfm import *  # noqa: F401,F403
from cdcmdr_amd.model import dfm as _m

globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
