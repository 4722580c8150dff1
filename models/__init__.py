"""Drop-in `models` package: the import lines of the reference drivers (mnist_moco_ode.py:5-6,
ucf_moco_ode.py:5-6) resolve to the MI355X-native classes of gan_ode_amd."""
