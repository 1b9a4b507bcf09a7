#!/usr/bin/env python3
"""Refresh profiles/pmc_latest.json from profiles/<tag>_pmc_fetch.txt / <tag>_pmc_write.txt (rounds 1-4: profiles/history/) (see its _method field).

    python tools/update_pmc_latest.py [tag]        # tag defaults to r01
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def grab(fn, kern, ctr):
    lines = open(fn).read().split("\n")
    for i, l in enumerate(lines):
        if kern in l:
            for m in lines[i + 1:i + 4]:
                if ctr in m:
                    return float(m.split()[1])
    raise SystemExit(f"{ctr} for {kern} not found in {fn}")


def grab_any(fn, kern, ctr):
    lines = open(fn).read().split("\n")
    for i, l in enumerate(lines):
        if kern in l:
            for m in lines[i + 1:i + 12]:
                if m.split() and m.split()[0] == ctr:
                    return float(m.split()[1])
    raise SystemExit(f"{ctr} for {kern} not found in {fn}")


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    p = os.path.join(ROOT, "profiles", "pmc_latest.json")
    d = json.load(open(p))
    for k, name in (("k_render_fwd", "k_render_fwd"), ("k_render_bwd", "k_render_bwd<true")):
        f = grab(os.path.join(ROOT, "profiles", f"{tag}_pmc_fetch.txt"), name, "FETCH_SIZE")
        w = grab(os.path.join(ROOT, "profiles", f"{tag}_pmc_write.txt"), name, "WRITE_SIZE")
        d[k].update(FETCH_SIZE_KiB=f, WRITE_SIZE_KiB=w, hbm_bytes=int((2 * f + w) * 1024))
        sq = os.path.join(ROOT, "profiles", f"{tag}_pmc_sq.txt")
        if os.path.exists(sq):
            # counters are averages per shader-engine instance (32 SEs x 32 SIMDs); ACTIVE_INST_VALU counts
            # quad-cycles, so VALU-busy = 4 * ACTIVE / (32 SIMDs * BUSY_CYCLES)
            act = grab_any(sq, name, "SQ_ACTIVE_INST_VALU")
            busy = grab_any(sq, name, "SQ_BUSY_CYCLES")
            insts = grab_any(sq, name, "SQ_INSTS_VALU")
            # VALU issue occupancy (the ACTIVE_INST formula read above 1 on the wide / tile kernels): every
            # wave-level VALU instruction holds its SIMD's issue port for at least 4 cycles (tools/valu_rate.hip: v_fma_f32 4.4,
            # v_pk_fma_f32 4.7, v_exp_f32 8.1), so  insts x 4 / (1024 SIMDs x busy cycles)  is a LOWER bound of the busy
            # fraction and cannot exceed 1; the old figure stays beside it as valu_active_quadcycles
            d[k].update(valu_busy=round(insts * 32.0 * 4.0 / (1024.0 * busy), 4), valu_insts=int(insts * 32),
                        valu_active_quadcycles=round(4.0 * act / (32.0 * busy), 4))
    # the x12, x8 and 16-per-LR-pixel legs of the bench line (same passes at those configs, when collected): HBM bytes per
    # launch of every kernel of the stage, {forward: .., backward: ..}; the tile-stationary backward counts its gather
    legs = {}
    for cfg in ("c3", "c4", "c2x16"):
        ff = os.path.join(ROOT, "profiles", f"{tag}_pmc_fetch_{cfg}.txt")
        fw = os.path.join(ROOT, "profiles", f"{tag}_pmc_write_{cfg}.txt")
        if not (os.path.exists(ff) and os.path.exists(fw)):
            continue

        def kernels(fn, ctr):
            out, name = {}, None
            for line in open(fn):
                t = line.split()
                if line.startswith("  ") and not line.startswith("      "):
                    name = line.strip()
                elif t and t[0] == ctr and name:
                    out[name] = float(t[1])
            return out
        fe, wr = kernels(ff, "FETCH_SIZE"), kernels(fw, "WRITE_SIZE")
        leg = {}
        for stage, keys in (("forward", ("k_render_fwd",)), ("backward", ("k_render_bwd", "k_bwd_gather"))):
            tot = 0.0
            for k in fe:
                if any(q in k for q in keys):
                    tot += (2 * fe[k] + wr.get(k, 0.0)) * 1024
            if tot:
                leg[stage] = int(tot)
        legs[cfg] = leg
    if legs:
        d["configs"] = legs
        d["_configs_source"] = f"profiles/{tag}_pmc_fetch_<config>.txt + {tag}_pmc_write_<config>.txt, same method; backward = render kernel + gather"
    d["_source"] = re.sub(r"r\d\d_", tag + "_", d["_source"])
    d["_valu_method"] = ("valu_busy = SQ_INSTS_VALU x 32 (SE instances) x 4 cycles / (1024 SIMDs x SQ_BUSY_CYCLES): a lower bound of the "
                         f"VALU issue occupancy that cannot exceed 1 (profiles/{tag}_pmc_sq.txt); valu_active_quadcycles = the rounds 1-4 "
                         "figure 4 * SQ_ACTIVE_INST_VALU / (32 * SQ_BUSY_CYCLES), uncalibrated (reads > 1 on some kernels); valu_insts = "
                         "wave-level VALU instructions per launch")
    try:
        import subprocess
        d["_build"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, text=True).strip()
    except Exception:
        pass
    d["_tag"] = tag
    json.dump(d, open(p, "w"), indent=1)
    print(json.dumps(d, indent=1))


if __name__ == "__main__":
    main()
