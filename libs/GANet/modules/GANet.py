"""Same import path as the reference (libs/GANet/modules/GANet.py); implementation: ganet_amd."""
from ganet_amd.modules.GANet import *  # noqa: F401,F403
from ganet_amd.modules.GANet import __all__  # noqa: F401
