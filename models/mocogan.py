"""Drop-in for the reference's models/mocogan.py: the discriminators the stage-3 ODE drivers import
(mnist_moco_ode.py:5), backed by libgode.so."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_ode_amd.modules import Noise, PatchImageDiscriminator, VideoDiscriminator  # noqa: E402,F401
