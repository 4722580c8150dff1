"""Drop-in for the reference's models/mocogan_ode_rnn.py (mnist_moco_ode_rnn.py:6)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_ode_amd.modules import ODEFunc, VideoGeneratorMNISTODERNN  # noqa: E402,F401
