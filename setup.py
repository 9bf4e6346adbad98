"""Build script behind pyproject.toml.

Two jobs: (1) copy the C-ABI header (include/pnr.h, at the repo root) into the installed package as
pointnerf2studio_amd/include/pnr.h, next to the HIP sources, so that an installed copy can build libpnr_hip.so on first
use; (2) with a setuptools older than 61 (no PEP 621: the `[project]` table of pyproject.toml is ignored) hand the same
metadata -- read from pyproject.toml, stated once -- to setup().  `python -m pointnerf2studio_amd.build` before packaging
puts the prebuilt library into the wheel.
"""
import os
import shutil

import setuptools
from setuptools import find_packages, setup
from setuptools.command.build_py import build_py

HERE = os.path.dirname(os.path.abspath(__file__))


class build_py_with_header(build_py):
    def run(self):
        super().run()
        src = os.path.join(HERE, "include", "pnr.h")
        if os.path.exists(src):
            dst = os.path.join(self.build_lib, "pointnerf2studio_amd", "include")
            os.makedirs(dst, exist_ok=True)
            shutil.copy2(src, os.path.join(dst, "pnr.h"))


def legacy_metadata():
    """The `[project]` table for a setuptools that does not read it."""
    try:
        import tomllib as toml
    except ImportError:
        import tomli as toml
    with open(os.path.join(HERE, "pyproject.toml"), "rb") as f:
        cfg = toml.load(f)
    prj, st = cfg["project"], cfg["tool"]["setuptools"]
    return dict(
        name=prj["name"], version=prj["version"], description=prj["description"],
        python_requires=prj["requires-python"], install_requires=prj["dependencies"],
        extras_require=prj.get("optional-dependencies", {}),
        packages=find_packages(where=HERE, include=st["packages"]["find"]["include"]),
        package_data=st["package-data"], include_package_data=False,
        entry_points={group: [f"{k} = {v}" for k, v in eps.items()] for group, eps in prj["entry-points"].items()},
    )


kwargs = {"cmdclass": {"build_py": build_py_with_header}}
if int(setuptools.__version__.split(".")[0]) < 61:
    kwargs.update(legacy_metadata())
setup(**kwargs)
