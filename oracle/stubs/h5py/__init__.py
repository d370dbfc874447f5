"""Minimal stand-in for `h5py`, used ONLY by oracle/make_goldens.py in the build container.

TEST INFRASTRUCTURE - never imported by the product path.

The reference's driver writes its result file through h5py (marlpde/Evolve_scenario.py:172-178).
The stand-in captures datasets and attributes in memory (``h5py.LAST`` holds the last file
written) so the unmodified driver can run where h5py is absent.
"""
import numpy as np

LAST = None


class File:
    def __init__(self, name, mode="r"):
        global LAST
        self.name = name
        self.mode = mode
        self.datasets = {}
        self.attrs = {}
        LAST = self

    def create_dataset(self, name, data=None, **_):
        self.datasets[name] = np.array(data)
        return self.datasets[name]

    def get(self, name):
        return self.datasets.get(name)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False
