from .GANet import *
