"""Builds, in this (GPU-less) container, the user-problem libraries the `-m gpu` tests load, so that they travel to the GPU
box with the snapshot instead of being compiled there; stale libraries of earlier source states are removed."""
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lpopc_amd import userproblem  # noqa: E402

HEADERS = ["tests/user_problems/bryson_denham_user.hpp", "tests/user_problems/brachistochrone_user.hpp",
           "examples/user_problem_vanderpol.hpp"]

if __name__ == "__main__":
    keep = {userproblem.build(os.path.join(ROOT, h)) for h in HEADERS}
    d = os.path.join(ROOT, "lpopc_amd", "csrc", "user_libs")
    for f in glob.glob(os.path.join(d, "librpm_hip_user_*.so")):
        if f not in keep:
            os.remove(f)
    tags = {os.path.basename(k)[len("librpm_hip_user_"):-3] for k in keep}
    for b in glob.glob(os.path.join(d, "build_*")):
        if os.path.basename(b)[len("build_"):] not in tags:
            shutil.rmtree(b)
    print("\n".join(sorted(keep)))
