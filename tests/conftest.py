import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_package():
    """import the hyphen-named package directory as `ddp_generator_amd`"""
    if "ddp_generator_amd" in sys.modules:
        return sys.modules["ddp_generator_amd"]
    path = os.path.join(ROOT, "ddp-generator_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location("ddp_generator_amd", path,
                                                  submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ddp_generator_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def pkg():
    return load_package()


@pytest.fixture(scope="session")
def oracle_built():
    """build the CPU checker (and, where /root/reference exists, the reference build)"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle", "ref"])
    return True


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def ref_available(full_ddp=0):
    from oracle.harness import lib_path
    return os.path.exists(lib_path("ref", full_ddp=full_ddp))
