"""Raw PMC text of scripts/prof_{conv,warp,conv3d}.sh (as scripts/collect_profiles.sh leaves it under
gpurun_out/collect/, or as published under profiles/rNN_*_pmc_raw.txt) -> profiles/rNN_pmc.json: HBM-side bytes PER
LAUNCH of the kernels bench.py prices (`roofline.traffic`, `roofline_hbm.traffic`, `also_3d.roofline.traffic`).

    python scripts/pmc_json.py r03                      # from gpurun_out/collect/pmc_*.txt
    python scripts/pmc_json.py r02 --from-profiles      # from profiles/r02_*_pmc_raw.txt

Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE
counts exactly half of the bytes of a 16-B/lane coalesced streaming read (128-B requests tallied at 64 B), so it is
doubled -- for EVERY kernel here, all of which stream their inputs with 16-B buffer / global loads.  WRITE_SIZE is
taken as is.  Values are per launch (the profile scripts average over the launches of a run)."""
import ast
import json
import os
import re
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(txt):
    dur, ctr = {}, {}
    for l in txt.replace("(anonymous namespace)::", "").splitlines():
        m = re.match(r'^(void )?([A-Za-z0-9_]+(?:<[^>]*>)?)\(.*? (\d+) ([\d.]+)$', l)
        if m:
            dur.setdefault(m.group(2), float(m.group(4)))
        m = re.match(r'^p\d (void )?([A-Za-z0-9_]+(?:<[^>]*>)?)\(.*?(\{.*\})$', l)
        if m:
            ctr.setdefault(m.group(2), {}).update(ast.literal_eval(m.group(3)))
    return dur, ctr


def entry(name, dur, c, alg_bytes, shape):
    f, w = c.get('FETCH_SIZE'), c.get('WRITE_SIZE')
    e = {"kernel": name, "shape": shape, "avg_launch_us": dur.get(name, 0.0) / 1e3, "algorithmic_bytes": alg_bytes,
         "fetch_kib_raw": f, "write_kib_raw": w}
    if f is not None and w is not None:
        e["fetch_bytes"] = 2.0 * f * 1024          # gfx950: FETCH_SIZE x 2 for 16-B/lane streams
        e["write_bytes"] = w * 1024.0
        e["traffic_bytes"] = e["fetch_bytes"] + e["write_bytes"]
        e["traffic_over_algorithmic"] = e["traffic_bytes"] / alg_bytes
    hit, req = c.get('TCC_HIT_sum'), c.get('TCC_REQ_sum')
    if hit is not None and req:
        e["l2_hit"] = hit / req
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c:
        cyc = c['GRBM_GUI_ACTIVE'] / 8.0
        e["mfma_busy"] = c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * cyc)
        e["lds_bank_conflict_cycles"] = c.get('SQ_LDS_BANK_CONFLICT')
    return e


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    from_prof = "--from-profiles" in sys.argv
    P = R + "/profiles/"
    C = R + "/gpurun_out/collect/"

    def rd(collect_names, prof_name, section=None):
        if from_prof:
            t = open(P + "%s_%s" % (tag, prof_name)).read()
            if section is not None:
                parts = re.split(r'^### .*$', t, flags=re.M)
                t = parts[section + 1] if len(parts) > section + 1 else t
            return t
        return "\n".join(open(C + n).read() for n in collect_names if os.path.exists(C + n))

    out = {"_note": "HBM-side bytes per launch from rocprofv3 --pmc passes (FETCH_SIZE x2 gfx950 correction applied to every "
                    "kernel, KiB -> bytes); regenerate with scripts/collect_profiles.sh + scripts/pmc_json.py " + tag}
    nv = 160 * 192 * 224
    # 2-D dominant shape: 256->256 3x3 @64^2, n = 32: x 134.2 MB + y 134.2 MB + split weights 3.5 MB
    d, c = parse(rd(["pmc_conv_fwd.txt", "pmc_conv_wgrad.txt"], "conv3x3s_pmc_raw.txt"))
    alg2d = 4.0 * 32 * 256 * 64 * 64 * 2 + 9 * 256 * 256 * 4 * 1.5
    for k in c:
        if 'split_cs_k' in k:
            out["conv3x3_split_cs_k"] = entry(k, d, c[k], alg2d, "256->256 3x3 @64x64, n=32 (154.6 GFLOP)")
        if 'wgrad_split2' in k:
            out["conv3x3_wgrad_split2_k"] = entry(k, d, c[k], alg2d, "256->256 3x3 @64x64, n=32 (154.6 GFLOP)")
    d, c = parse(rd(["pmc_warp.txt"], "warp_pmc_raw.txt"))
    for k in c:
        if 'warp_win_fwd_k' in k:
            out["warp_win_fwd_k"] = entry(k, d, c[k], 4.0 * 5 * nv, "160x192x224, C=1, smooth field")
    own = [k for k in c if 'bwd_own' in k or 'warp_win_gather' in k or 'warp_win_slow' in k or 'bwd_fx' in k]
    if own and all('FETCH_SIZE' in c[k] and 'WRITE_SIZE' in c[k] for k in own):
        tot = {"kernel": " + ".join(own), "shape": "160x192x224, C=1, smooth field: d(src) + d(flow)",
               "algorithmic_bytes": 4.0 * 9 * nv,
               "avg_launch_us": sum(d.get(k, 0.0) for k in own) / 1e3,
               "fetch_bytes": sum(2.0 * c[k]['FETCH_SIZE'] * 1024 for k in own),
               "write_bytes": sum(c[k]['WRITE_SIZE'] * 1024.0 for k in own)}
        tot["traffic_bytes"] = tot["fetch_bytes"] + tot["write_bytes"]
        tot["traffic_over_algorithmic"] = tot["traffic_bytes"] / tot["algorithmic_bytes"]
        out["warp_bwd_dsrc_dflow"] = tot
    for sec, (cin, cout) in enumerate(((34, 32), (32, 16))):
        d, c = parse(rd(["pmc_conv3d_%d_%d.txt" % (cin, cout)], "conv3d_pmc_raw.txt", section=sec))
        alg = 4.0 * nv * (cin + cout) + 27 * cin * cout * 4 * 1.5
        for k in c:
            if 'conv3d_split_k' in k and 'FETCH_SIZE' in c[k]:
                out["conv3d_split_k_%d_%d" % (cin, cout)] = entry(k, d, c[k], alg, "%d->%d 3x3x3 @160x192x224" % (cin, cout))
            if 'conv3d_split_m16_k' in k and 'FETCH_SIZE' in c[k]:      # <= 16 output channels: the 16-row MFMA form
                out["conv3d_split_m16_k_%d_%d" % (cin, cout)] = entry(k, d, c[k], alg, "%d->%d 3x3x3 @160x192x224" % (cin, cout))
            if 'wgrad_tr' in k and 'FETCH_SIZE' in c[k]:
                out["conv3d_wgrad_tr_k_%d_%d" % (cin, cout)] = entry(k, d, c[k], alg, "%d->%d 3x3x3 @160x192x224" % (cin, cout))
            if 'conv3d_march_k' in k and 'FETCH_SIZE' in c[k]:          # the z-marching kernel (csrc/conv3dm.hip)
                out["conv3d_march_k_%d_%d" % (cin, cout)] = entry(k, d, c[k], alg, "%d->%d 3x3x3 @160x192x224" % (cin, cout))
            if 'conv3d_wgrad_march_k' in k and 'FETCH_SIZE' in c[k]:    # the marching weight gradient (csrc/conv3dwm.hip)
                out["conv3d_wgrad_march_k_%d_%d" % (cin, cout)] = entry(k, d, c[k], alg, "%d->%d 3x3x3 weight gradient @160x192x224" % (cin, cout))
    if not from_prof and os.path.exists(C + "pmc_conv3d_march.txt"):
        d, c = parse(open(C + "pmc_conv3d_march.txt").read())
        for k in c:
            m_ = re.match(r'conv3d_march_k<(\d+), (\d+)', k)
            if m_ and 'FETCH_SIZE' in c[k]:
                cin, cout = int(m_.group(1)), int(m_.group(2))
                out["conv3d_march_k_%d_%d" % (cin, cout)] = entry(k, d, c[k], 4.0 * nv * (cin + cout) + 27 * cin * cout * 4,
                                                                  "%d->%d 3x3x3 @160x192x224" % (cin, cout))
    if not from_prof and os.path.exists(C + "pmc_upconv3d.txt"):
        d, c = parse(open(C + "pmc_upconv3d.txt").read())
        lo = 80 * 96 * 112
        for k in c:
            if 'conv3d_up_phase_k<true>' in k and 'FETCH_SIZE' in c[k]:
                out["conv3d_up_phase_k"] = entry(k, d, c[k], 4.0 * (32 * lo + 2 * nv + 32 * nv),
                                                 "32 (half res) + 2 -> 32 @160x192x224, one launch (404 GFLOP reference-equivalent)")
            if 'conv3d_up_dgrad_k' in k and 'FETCH_SIZE' in c[k]:
                out["conv3d_up_dgrad_k"] = entry(k, d, c[k], 4.0 * (32 * nv + 32 * lo), "d(a): 32 @160x192x224 -> 32 @80x96x112")
            if 'conv3d_upwgrad4_k' in k and 'FETCH_SIZE' in c[k]:      # csrc/conv3duw.hip: up-sampled channels in parity classes
                out["conv3d_upwgrad4_k"] = entry(k, d, c[k], 4.0 * (32 * lo + 2 * nv + 32 * nv),
                                                 "34 -> 32 weight gradient: cat(up2(a), b) in parity classes, skip channels + db fused")
            if 'wgrad_tr_k<3, false, true>' in k and 'FETCH_SIZE' in c[k]:
                out["conv3d_wgrad_tr_k_upcat"] = entry(k, d, c[k], 4.0 * (32 * lo + 2 * nv + 32 * nv), "34 -> 32 weight gradient, operand cat(up2(a), b) read in place")
    json.dump(out, open(P + tag + "_pmc.json", "w"), indent=1, sort_keys=True)
    print("wrote", P + tag + "_pmc.json", sorted(k for k in out if not k.startswith("_")))


if __name__ == "__main__":
    main()
