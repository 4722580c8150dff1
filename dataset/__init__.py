"""Drop-in `dataset` package for the reference drivers' `from dataset import MNISTRotationVideo, MNISTRotationImage`
(mnist_moco_ode.py:4).  Only the Rotated-MNIST classes on the hot path's input side are provided; the UCF101 video
decoding stack (PyAV / torchvision) is out of scope."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_ode_amd.data import MNISTRotationImage, MNISTRotationVideo  # noqa: E402,F401
