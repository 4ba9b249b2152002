#!/usr/bin/env python3
"""tools/stamp.py — writes build/STAMP.json in the build container, before a gpurun call that collects evidence: the git revision
the tree was built from (the GPU box gets a snapshot without .git), whether the tree differs from it, and the sha256 of
lib/libssd_hip.so.  tools/r06_final.sh (on the GPU box) checks the library it runs against this stamp and copies it into every
counter file it makes; bench.py compares the stamp of the committed counters with the library it loaded (`stale`)."""
import hashlib
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for block in iter(lambda: f.read(1 << 20), b""):
            h.update(block)
    return h.hexdigest()


def main():
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "stair-step-detector_amd/csrc", "include"], capture_output=True, text=True).stdout.strip()
    stamp = {"git_head": head, "source_tree_differs_from_head": bool(dirty),
             "lib_sha256": sha256(os.path.join(ROOT, "stair-step-detector_amd", "lib", "libssd_hip.so"))}
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    json.dump(stamp, open(os.path.join(ROOT, "build", "STAMP.json"), "w"), indent=1)
    print(json.dumps(stamp))


if __name__ == "__main__":
    main()
