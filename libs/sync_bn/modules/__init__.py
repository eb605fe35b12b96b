from .sync_bn import *
