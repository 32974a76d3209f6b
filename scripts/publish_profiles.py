"""Copy gpurun_out/collect/* (scripts/collect_profiles.sh on a GPU box) into profiles/<TAG>_* and regenerate the
PMC summaries (<TAG>_conv3x3s_pmc.md, <TAG>_conv3d_pmc.md, <TAG>_conv3dup_pmc.md) from the raw counter output.
usage: python scripts/publish_profiles.py [r04]"""
import ast, os, re, sys
TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C, P = R + '/gpurun_out/collect/', R + '/profiles/'
cp = {'bench.json': TAG + '_bench_b16.json', 'bench_eager.json': TAG + '_bench_b16_eager.json', 'bench_kernel_stats.csv': TAG + '_bench_b16_kernel_stats.csv',
      'step_trace.txt': TAG + '_step_trace.txt', 'launch_census.txt': TAG + '_launch_census.txt', 'bench_conv.txt': TAG + '_bench_conv.txt',
      'bench_conv3d.txt': TAG + '_bench_conv3d.txt', 'bench_conv3d_fp32.txt': TAG + '_bench_conv3d_fp32.txt', 'bench_3d.txt': TAG + '_bench_3d.txt',
      'bench_3d_kernel_stats.csv': TAG + '_bench_3d_kernel_stats.csv', 'bench_hbm.txt': TAG + '_bench_hbm.txt',
      'bench_warp_roofline.json': TAG + '_bench_warp_roofline.json', 'bench_wgrad3d.txt': TAG + '_bench_wgrad3d.txt',
      'step_trace_3d.txt': TAG + '_step_trace_3d.txt', 'sustain3d.txt': TAG + '_power_clock_3d.txt', 'pmc_warp.txt': TAG + '_warp_pmc_raw.txt',
      'step_trace_3d_128.txt': TAG + '_step_trace_3d_128.txt', 'bench_upconv3d.txt': TAG + '_bench_upconv3d.txt',
      'ab_xcd_order.txt': TAG + '_ab_xcd_order.txt', 'overlap_trace.txt': TAG + '_overlap_trace.txt', 'bench_in_blurdown.txt': TAG + '_bench_in_blurdown.txt',
      'conv2d_layer_census.txt': TAG + '_conv2d_layer_census.txt', 'conv3d_step_census.txt': TAG + '_conv3d_step_census.txt', 'ab_round3_switches.txt': TAG + '_ab_round3_switches.txt', 'ab_switches.txt': TAG + '_ab_switches.txt', 'pmc_upconv3d.txt': TAG + '_conv3dup_pmc_raw.txt', 'pmc.json': TAG + '_pmc.json', 'bench_march.txt': TAG + '_bench_march.txt', 'ab_march_3d.txt': TAG + '_ab_march_3d.txt', 'march_trace.txt': TAG + '_march_trace.txt', 'bench_upwgrad.txt': TAG + '_bench_upwgrad.txt', 'upwgrad_ko.txt': TAG + '_upwgrad_ko.txt',
      'ab_upwgrad_3d.txt': TAG + '_ab_upwgrad_3d.txt', 'ab_wgrad_march_3d.txt': TAG + '_ab_wgrad_march_3d.txt',
      'ab_staged.txt': TAG + '_ab_staged.txt', 'ab_1x1_wgrad.txt': TAG + '_ab_1x1_wgrad.txt', 'ab_deterministic.txt': TAG + '_ab_deterministic.txt',
      'ab_flow_wgrad.txt': TAG + '_ab_flow_wgrad.txt', 'graph_split_probe.txt': TAG + '_graph_split_probe.txt', 'parity_margins.txt': TAG + '_parity_margins.txt', 'bench_conv1x1.txt': TAG + '_bench_conv1x1.txt',
      'pmc_conv1x1.txt': TAG + '_conv1x1_pmc_raw.txt', 'pmc_flow_wgrad.txt': TAG + '_flow_wgrad_pmc_raw.txt',
      'ab_3d_round6.txt': TAG + '_ab_3d_round6.txt', 'bench_flow_head.txt': TAG + '_bench_flow_head.txt', 'pmc_step3d.txt': TAG + '_step3d_pmc_raw.txt', 'ab_smooth.txt': TAG + '_ab_smooth.txt'}
def clean(txt):
    txt = txt.replace("(anonymous namespace)::", "")
    return "\n".join(l for l in txt.splitlines() if not re.match(r'^[WEI]\d{8} ', l) and 'amdgpu.ids' not in l and 'UserWarning' not in l and '_warn_once' not in l)
for a, b in cp.items():
    if not os.path.exists(C + a):
        continue
    t = open(C + a).read()
    if a.endswith('.txt'):
        t = clean(t) + "\n"
    open(P + b, 'w').write(t)
raw2d = clean(open(C + 'pmc_conv_fwd.txt').read()) + "\n" + clean(open(C + 'pmc_conv_wgrad.txt').read()) + "\n"
open(P + TAG + '_conv3x3s_pmc_raw.txt', 'w').write(raw2d)
s1, s2 = clean(open(C + 'pmc_conv3d_34_32.txt').read()), clean(open(C + 'pmc_conv3d_32_16.txt').read())
s3 = clean(open(C + 'pmc_conv3d_march.txt').read()) if os.path.exists(C + 'pmc_conv3d_march.txt') else ""
open(P + TAG + '_conv3d_pmc_raw.txt', 'w').write("### scripts/prof_conv3d.sh 34-32\n" + s1 + "\n### scripts/prof_conv3d.sh 32-16\n" + s2 + "\n### scripts/prof_conv3d.sh 16-16,16-32\n" + s3 + "\n")
def parse(txt):
    dur, ctr = {}, {}
    for l in txt.splitlines():
        m = re.match(r'^(void )?([A-Za-z0-9_]+(?:<[^>]*>)?)\(.*? (\d+) ([\d.]+)$', l)
        if m: dur[m.group(2)] = float(m.group(4)) / 1e3
        m = re.match(r'^p\d (void )?([A-Za-z0-9_]+(?:<[^>]*>)?)\(.*?(\{.*\})$', l)
        if m: ctr.setdefault(m.group(2), {}).update(ast.literal_eval(m.group(3)))
    return dur, ctr
def row(name, dur, c, gflop, note=""):
    cyc = c['GRBM_GUI_ACTIVE'] / 8
    busy = c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc)
    return "| `%s` | %.0f µs → %.0f TF | %.0f k cycles → %.2f GHz | **%.1f %%** | %.1f M (%.0f %% of the LDS cycles; conflicts %.1f M) | fetch %.0f MB, write %.0f MB; L2 hit %.0f %% | %s |" % (
        name, dur, gflop / dur * 1e3, cyc / 1e3, cyc / dur / 1e3, 100 * busy, c['SQ_LDS_IDX_ACTIVE'] / 1e6,
        100 * c['SQ_LDS_IDX_ACTIVE'] / 256 / cyc, c['SQ_LDS_BANK_CONFLICT'] / 1e6, c['FETCH_SIZE'] * 1024 / 1e6,
        c['WRITE_SIZE'] * 1024 / 1e6, 100 * c['TCC_HIT_sum'] / max(c['TCC_REQ_sum'], 1), note)
hdr = "| kernel | duration → algorithmic rate | clock (GRBM_GUI_ACTIVE/8 ÷ t) | matrix pipe busy: SQ_VALU_MFMA_BUSY_CYCLES ÷ (1024 SIMD × cycles) | LDS active | HBM side | note |\n|---|---|---|---|---|---|---|\n"
d, c = parse(raw2d)
k1 = [k for k in c if 'split_cs_k' in k][0]; k2 = [k for k in c if 'wgrad_split2' in k][0]
md = ("# PMC counters of the split 3×3 kernels on this round's binary (`csrc/conv3x3s.hip`), re-taken with the final profile run\n\n"
      "Command: `scripts/prof_conv.sh {fwd|wgrad} 256 256 64 32` — one counter group per pass, `--kernel-trace` only (MI355X_MICROARCH.md): 256→256 3×3 reflect @64², n = 32, **154.6 GFLOP** algorithmic per launch.  Raw: `{TAG}_conv3x3s_pmc_raw.txt`.  Counters summed over the 8 XCDs; FETCH/WRITE_SIZE in KiB.\n\n" + hdr)
md += row(k1, d[k1], c[k1], 154.6, "non-MFMA VALU per wave and 16-channel chunk: %.0f" % ((c[k1]['SQ_INSTS_VALU'] - c[k1]['SQ_VALU_MFMA_BUSY_CYCLES'] / 32) / (8192 * 16))) + "\n"
md += row(k2, d[k2], c[k2], 154.6, "non-MFMA VALU per wave and run: %.0f; FETCH of its 16-B/lane streams under-counts by 2 (guide's correction)" % ((c[k2]['SQ_INSTS_VALU'] - c[k2]['SQ_VALU_MFMA_BUSY_CYCLES'] / 32) / (2048 * 128))) + "\n"
md += "\nBoth kernels sit on the 1 400 W package cap (`r01_power_clock.md`); the busy fraction moves with the clock the box sustains (round 1: 68.0 % / 56.8 % at 1.72 / 1.84 GHz).  `bench.py` prices the same kernels over the step's shape mix as `roofline.issued_frac` / `wgrad_issued_frac` (3 products per MAC over the 2.5 PFLOP/s dense fp16 peak; `frac` is the algorithmic third of it) -- see the committed `{TAG}_bench_b16.json`.\n"
open(P + TAG + '_conv3x3s_pmc.md', 'w').write(md.replace('{TAG}', TAG))
md3 = ("# PMC counters of the 3-D split kernels (`csrc/conv3ds.hip`, `csrc/conv3dm.hip`), this round's final binary\n\n"
       "Command: `scripts/prof_conv3d.sh 34-32` and `… 32-16` (`scripts/bench_conv3d.py` under `rocprofv3 --kernel-trace --pmc <group>`, one group per pass): 160×192×224, 404.3 GFLOP (34→32) / 190.3 GFLOP (32→16) per launch.  Raw: `{TAG}_conv3d_pmc_raw.txt`.  The clock column is what the counters give for the profiler's serialised single launches between other work; the sustained figures are below.\n\n" + hdr)
def gf_of(k, gf):
    m = re.match(r'conv3d_march_k<(\d+), (\d+)', k)
    return 2.0 * int(m.group(1)) * int(m.group(2)) * 27 * 160 * 192 * 224 / 1e9 if m else gf
for sec, gf in ((s1, 404.29), (s2, 190.25), (s3, 95.13)):
    d, c = parse(sec)
    for k in c:
        if k in d and ('conv3d_split_k' in k or 'wgrad_tr' in k or 'conv3d_march_k' in k or 'split_m16' in k or 'wgrad_march' in k) and 'FETCH_SIZE' in c[k] and 'SQ_LDS_IDX_ACTIVE' in c[k] and 'GRBM_GUI_ACTIVE' in c[k]:
            if sec is s3 and 'conv3d_march_k' not in k:
                continue
            md3 += row(k, d[k], c[k], gf_of(k, gf), ("marching weight gradient, 27 tap matrices resident (csrc/conv3dwm.hip)" if 'wgrad_march' in k else ("z-marching kernel (csrc/conv3dm.hip)" if 'march' in k else ""))) + "\n"
md3 += ("\nSustained clocks / package power with each kernel running back to back (`scripts/sustain_clock3d.py`, `{TAG}_power_clock_3d.txt`): the weight-gradient kernel holds the package AT its 1 400 W cap (1.72–1.81 GHz), the forward kernel just under it (1 377–1 392 W at 1.98–2.04 GHz).  On the cap only energy per useful FLOP buys speed: skipping the padding row tile (1/8 of the reads and MFMAs of the 7-tile form) took 34→32 from 1.815 to 1.715 ms although no wave finishes earlier; making the forward kernel persistent, prefetching across tiles or staggering the two workgroups of a CU changed nothing.\n")
open(P + TAG + '_conv3d_pmc.md', 'w').write(md3.replace('{TAG}', TAG))
if os.path.exists(C + 'pmc_upconv3d.txt'):
    su = clean(open(C + 'pmc_upconv3d.txt').read())
    d, c = parse(su)
    mdu = ("# PMC counters of the parity-class kernels of the nearest_up2 + cat layers (`csrc/conv3ds.hip`, `csrc/conv3duw.hip`)\n\n"
           "Command: `scripts/prof_upconv3d.sh` (`scripts/bench_upconv3d.py`: a [1,32,80,96,112], b [1,2,160,192,224], 34 -> 32 channels, "
           "404.3 GFLOP reference-equivalent per pass; the kernels execute 8/27 of the products of the 32 up-sampled channels).  Raw: `"
           + TAG + "_conv3dup_pmc_raw.txt`.  Rates in the table are REFERENCE-EQUIVALENT FLOP/s.\n\n" + hdr)
    for k in c:
        if k in d and ('up_phase' in k or 'up_dgrad' in k or 'wgrad_tr' in k or 'upwgrad4' in k) and 'GRBM_GUI_ACTIVE' in c[k] and 'FETCH_SIZE' in c[k] and 'SQ_LDS_IDX_ACTIVE' in c[k]:
            gf = 380.5 if 'up_dgrad' in k else 404.29
            mdu += row(k, d[k], c[k], gf) + "\n"
    open(P + TAG + '_conv3dup_pmc.md', 'w').write(mdu)
print("published")
