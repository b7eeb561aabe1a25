"""Packaging of rri_nmf_amd.  The HIP library is built IN-TREE (rri_nmf_amd/lib/librri_hip.so) by
`python -m rri_nmf_amd.build` (hipcc, --offload-arch=gfx950); `build_py` / `develop` run that step first when hipcc is
on the PATH, and the .so ships as package data.  There is no CPU build: without the library the package imports but
every device entry point raises RRIHipUnavailable."""
import os
import shutil
import sys

from setuptools import setup
from setuptools.command.build_py import build_py
from setuptools.command.develop import develop

HERE = os.path.dirname(os.path.abspath(__file__))


def build_hip_library():
    if shutil.which('hipcc') is None and not os.path.exists('/opt/rocm/bin/hipcc'):
        sys.stderr.write('setup.py: hipcc not found, librri_hip.so not built (python -m rri_nmf_amd.build does it later)\n')
        return
    sys.path.insert(0, HERE)
    from rri_nmf_amd import build as hip_build
    hip_build.build()


class BuildPyWithHip(build_py):
    def run(self):
        build_hip_library()
        build_py.run(self)


class DevelopWithHip(develop):
    def run(self):
        build_hip_library()
        develop.run(self)


setup(
    name='rri_nmf_amd',
    version='0.1.0',
    description='Rank-one residue iteration NMF on AMD MI355X: HIP kernels behind the nmf() / estimator surface of rri_nmf',
    packages=['rri_nmf_amd'],
    package_data={'rri_nmf_amd': ['lib/*.so', 'csrc/*.hip', 'csrc/*.hpp']},
    data_files=[('include', ['include/rri_hip.h'])],
    python_requires='>=3.8',
    install_requires=['numpy', 'scipy', 'scikit-learn'],
    cmdclass={'build_py': BuildPyWithHip, 'develop': DevelopWithHip},
)
