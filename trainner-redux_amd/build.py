"""Builds csrc/*.hip into libtrainner_hip.so (in-tree, next to this file) with hipcc for gfx950."""
from __future__ import annotations

import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libtrainner_hip.so")
RESOURCES = os.path.join(HERE, "kernel_resources.json")   # per-kernel VGPRs / scratch / occupancy from the compiler (tests assert scratch == 0)
SOURCES = ["common.hip", "conv3x3.hip", "conv_ring.hip", "pack.hip", "wgrad3x3.hip", "wgrad_ring.hip", "eltwise.hip", "loss.hip", "optim.hip", "stem.hip", "disc.hip", "token.hip", "attn_mfma.hip", "metrics.hip", "gated.hip", "otf.hip", "winograd.hip", "winograd43.hip", "bnorm.hip"]


def _stale() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(RESOURCES):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(INCLUDE, "trainner_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def _parse_resources(text: str) -> dict:
    """`-Rpass-analysis=kernel-resource-usage` remarks -> {demangled-ish kernel name: {vgprs, agprs, scratch, occupancy, lds}}."""
    out, cur = {}, None
    keys = {"VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occupancy",
            "LDS Size [bytes/block]": "lds", "VGPRs Spill": "vgpr_spill"}
    for ln in text.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", ln)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", ln)
        if m and cur is not None and m.group(1).strip() in keys:
            cur[keys[m.group(1).strip()]] = int(m.group(2))
    return out


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in SOURCES:
        o = os.path.join(HERE, "build", s.replace(".hip", ".o"))
        objs.append(o)
        dev = ["-DTR_RING_DEV"] + (["-DTR_RING_STORE_AUX=" + os.environ["TR_AUX"]] if os.environ.get("TR_AUX") else []) if os.environ.get("TRAINNER_DEV_BUILD") == "1" else []   # kernel development: fewer template instances
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", *dev, "-Rpass-analysis=kernel-resource-usage", f"-I{INCLUDE}",
               f"-I{CSRC}", "-c", os.path.join(CSRC, s), "-o", o]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    resources = {}
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
        resources.update(_parse_resources(out))
        rest = "\n".join(ln for ln in out.splitlines() if "kernel-resource-usage" not in ln and "remark:" not in ln)
        if verbose and rest.strip():
            print(rest)
    with open(RESOURCES, "w") as f:
        json.dump(resources, f, indent=1, sort_keys=True)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
