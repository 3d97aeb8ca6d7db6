"""The boundary from plain C: examples/c_abi_example.c (gcc + dlopen, no Python in the call path) drives
glrm_*_create / fit / objective / destroy.  CPU: against the oracle library; -m gpu: against libglrm_hip.so, and both
trajectories agree."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build", "c_abi_example")


def build_example():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    subprocess.run(["gcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_example.c"), "-o", EXE,
                    "-ldl"], check=True)


def run_example(lib, prefix):
    r = subprocess.run([EXE, lib, prefix], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"(\d+) iterations, objective ([\d.eE+-]+) -> ([\d.eE+-]+) \(loss\+ry\), full objective ([\d.eE+-]+)", r.stdout)
    assert m, r.stdout
    return int(m.group(1)), float(m.group(2)), float(m.group(3)), float(m.group(4))


MULTI_EXE = os.path.join(ROOT, "build", "c_abi_multi")


def build_multi_example():
    os.makedirs(os.path.dirname(MULTI_EXE), exist_ok=True)
    subprocess.run(["gcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_multi.c"), "-o", MULTI_EXE,
                    "-ldl"], check=True)


def run_multi_example(lib, prefix, shards, devices=None):
    cmd = [MULTI_EXE, lib, prefix, str(shards)] + ([devices] if devices else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "BIT-IDENTICAL" in r.stdout, r.stdout
    return r.stdout


def oracle_lib_path():
    import oracle as O
    return O.build_oracle()


def test_c_program_drives_the_oracle_library():
    build_example()
    it, o0, o1, full = run_example(oracle_lib_path(), "glrm_cpu_")
    assert it >= 11 and o1 < o0 and full >= o1


def test_c_program_fails_loudly_without_gpu_or_library():
    build_example()
    r = subprocess.run([EXE, os.path.join(ROOT, "does_not_exist.so")], capture_output=True, text=True)
    assert r.returncode == 2 and "dlopen" in r.stderr


@pytest.mark.gpu
def test_c_program_drives_the_hip_library_and_matches_the_oracle():
    build_example()
    a = run_example(os.path.join(ROOT, "lowrankmodels.jl_amd", "libglrm_hip.so"), "glrm_hip_")
    b = run_example(oracle_lib_path(), "glrm_cpu_")
    assert a[0] == b[0]
    for x, y in zip(a[1:], b[1:]):
        assert x == pytest.approx(y, rel=1e-5)


@pytest.mark.parametrize("shards", [1, 2, 5])
def test_c_program_drives_the_multi_shard_entry_points_of_the_oracle(shards):
    """examples/c_abi_multi.c: glrm_*_multi_create / multi_fit / multi_info / multi_destroy from plain C, N shards in one process."""
    build_multi_example()
    out = run_multi_example(oracle_lib_path(), "glrm_cpu_", shards)
    assert f"{shards} shards" in out


@pytest.mark.gpu
@pytest.mark.parametrize("shards", [2, 4])
def test_c_program_drives_the_multi_gpu_entry_points_on_one_device(shards):
    """Two / four shards of ONE process on device 0 (own streams, replicas and copy streams per shard; the peer pushes become
    device-to-device copies): bit-identical to the single-shard fit."""
    build_multi_example()
    out = run_multi_example(os.path.join(ROOT, "lowrankmodels.jl_amd", "libglrm_hip.so"), "glrm_hip_", shards, ",".join(["0"] * shards))
    assert "exchange direct" in out
