#!/usr/bin/env python3
"""Instruction census of the device code of one stages translation unit (round 6: typed address spaces on the traversal path).

  python scripts/isa_census.py [stages.hip|stages_lat.hip|...] [extra hipcc flags...]

Compiles the unit device-only to gfx950 assembly with the product flags (restir_amd/build.py) and prints, per kernel: flat_ / global_ / ds_ / scratch_ / buffer_
loads and stores, s_waitcnt that wait on BOTH counters (vmcnt and lgkmcnt) at once, VGPRs, spills, scratch bytes.  Runs on the CPU box (hipcc cross-compiles)."""
import os, re, subprocess, sys, tempfile, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cis-565-final-vr-raytracer_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-Os", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize", "-Wno-unused-result",
         "-Wno-unused-command-line-argument", "-x", "hip", "--offload-device-only", "-S"]

def census(src="stages.hip", extra=(), keep=None):
    out = keep or os.path.join(tempfile.mkdtemp(prefix="isa_"), "unit.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + list(extra) + [os.path.join(CSRC, src), "-o", out])
    kern, rows, meta = None, collections.OrderedDict(), {}
    for line in open(out):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m and "k_" in m.group(1):
            kern = m.group(1); rows.setdefault(kern, collections.Counter()); continue
        if kern is None: continue
        s = line.strip()
        m = re.match(r"^(flat|global|ds|scratch|buffer)_(load|store|read|write|atomic)\w*", s)
        if m:
            rows[kern][m.group(1) + "_" + ("ld" if m.group(2) in ("load", "read") else ("st" if m.group(2) in ("store", "write") else "at"))] += 1
            if s.startswith("flat_load_dwordx2"): rows[kern]["flat_ld_x2"] += 1
            if s.startswith("ds_read_b64") or s.startswith("ds_write_b64"): rows[kern]["ds_b64"] += 1
            if "Folded Spill" in s: rows[kern]["spill_st"] += 1
        elif s.startswith("s_waitcnt"):
            rows[kern]["waitcnt"] += 1
            if "vmcnt" in s and "lgkmcnt" in s: rows[kern]["wait_both"] += 1
        elif re.match(r"^[vs]_\w+", s):
            rows[kern]["valu" if s.startswith("v_") else "salu"] += 1
        m = re.match(r"^;\s*(NumVgprs|ScratchSize|Occupancy|NumSgprs|LDSByteSize)\W+(\d+)", s, re.I)
        if m: meta.setdefault(kern, {})[m.group(1)] = int(m.group(2))
    return rows, meta, out

def short(k):
    m = re.search(r"\d+(k_[a-z_]+?)(ENS|E8|EP|I(L[b01E]+)E)", k)
    if not m: return k
    return m.group(1) + ("<" + m.group(3).replace("Lb", "").replace("E", "") + ">" if m.group(3) else "")

if __name__ == "__main__":
    src = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".hip") else "stages.hip"
    extra = [a for a in sys.argv[1:] if not a.endswith(".hip")]
    rows, meta, path = census(src, extra)
    cols = ["flat_ld", "flat_ld_x2", "flat_st", "global_ld", "global_st", "ds_ld", "ds_st", "ds_b64", "scratch_ld", "scratch_st", "waitcnt", "wait_both", "valu", "salu"]
    print("# %s %s -> %s" % (src, " ".join(extra), path))
    print("%-46s " % "kernel" + " ".join("%10s" % c for c in cols) + "   vgpr spill scratchB")
    for k, c in rows.items():
        if not any(c.values()): continue
        mt = meta.get(k, {})
        print("%-46s " % short(k)[:46] + " ".join("%10d" % c[x] for x in cols) + "   %4s %5s %6s" % (mt.get("NumVgprs", "?"), c["spill_st"], mt.get("ScratchSize", "?")))
