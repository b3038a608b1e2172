# SPDX-License-Identifier: GPL-3.0-or-later
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


# What a fresh GPU box executes first are the tests that pin the HIP path to the reference's own vectors and to the
# BASELINE configurations at full size; the randomised sweeps follow, and the tests that start other programs (the C++
# facade, the command line, several ranks on one GPU, bench.py) come last: under `-x` an environment-sensitive failure
# in one of those then costs nothing in front of it.  Files not listed keep their alphabetical place ahead of these.
GPU_FILE_ORDER = ["test_gpu_parity.py", "test_gpu_full_size.py", "test_gpu_tail_groups.py", "test_gpu_fuzz.py",
                  "test_gpu_health.py", "test_facade.py", "test_result_utils.py", "test_gpu_multi.py", "test_gpu_bench.py"]


def pytest_collection_modifyitems(session, config, items):
    rank = {name: i for i, name in enumerate(GPU_FILE_ORDER)}
    items.sort(key=lambda it: rank.get(os.path.basename(str(it.fspath)), -1))   # stable: order inside a file is kept


def load_golden(name):
    with open(os.path.join(HERE, "golden", name), encoding="utf-8") as f:
        return json.load(f)


def load_tiny():
    """tests/golden/diff_tiny.json.gz (oracle/gen_tiny_golden.py) -> (search cases, engine cases) in the
    field names of the other golden files; inputs are decoded to numpy arrays."""
    import base64
    import gzip
    import numpy as np
    with gzip.open(os.path.join(HERE, "golden", "diff_tiny.json.gz"), "rb") as f:
        doc = json.loads(f.read().decode())
    search = [dict(elem_bytes=c["e"], keyword=c["k"], wildcard=c["w"], char_seq=c["s"], values=c["v"],
                   data=np.frombuffer(base64.b64decode(c["d"]), dtype=np.uint8 if c["e"] == 1 else "<u2"), expect=c["x"])
              for c in doc["search"]]
    engine = [dict(elem_bytes=c["e"], keyword=c["k"], wildcard=c["w"], big_endian=c["be"], block_size=c["b"],
                   file=np.frombuffer(base64.b64decode(c["f"]), dtype=np.uint8), expect=c["x"]) for c in doc["engine"]]
    return search, engine


def build_fake_rccl():
    """tests/shim/libfake_rccl.so: the TEST-ONLY stand-in for the RCCL entry points the library calls, so that two ranks
    can share the one GPU a test box has (tests/shim/fake_rccl.cpp).  Plain g++: it links against neither HIP nor RCCL."""
    import subprocess
    src = os.path.join(HERE, "shim", "fake_rccl.cpp")
    so = os.path.join(HERE, "shim", "libfake_rccl.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-I" + os.path.join(rocm, "include"), src, "-o", so,
                               "-ldl", "-lrt", "-pthread"])
    return so


@pytest.fixture(scope="session")
def oracle():
    from _oracle import Oracle
    return Oracle()


def load_package():
    """Import monkey-moore_amd/ (hyphenated directory) as module `monkey_moore_amd`."""
    import importlib.util
    if "monkey_moore_amd" in sys.modules:
        return sys.modules["monkey_moore_amd"]
    pkg_dir = os.path.join(ROOT, "monkey-moore_amd")
    spec = importlib.util.spec_from_file_location(
        "monkey_moore_amd", os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["monkey_moore_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def mm():
    mod = load_package()
    mod.build.build_all()
    return mod


@pytest.fixture(scope="session")
def gpu_engine(mm):
    if mm.device_count() == 0:
        pytest.fail("no HIP device: the gpu tests need a real MI355X (there is no CPU fallback)")
    eng = mm.Engine(0)
    yield eng
    health = eng.health()
    eng.close()
    # whatever the session's scans went through: no published block may have failed the library's validation
    # (a rerun through the plain kernels heals the result, but the suite wants to know)
    assert health["fallback_reason"] == 0 and health["selftest"] in (0, 1), health
