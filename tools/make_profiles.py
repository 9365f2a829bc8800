"""Condense gpurun_out/<run>/ (rocprofv3 --stats, --pmc passes, bench JSON, tuner log) into profiles/<tag>_*."""
import collections, csv, glob, json, os, re, shutil, sys
src, tag = sys.argv[1], sys.argv[2]
os.makedirs("profiles", exist_ok=True)


def short(n):
    n = re.sub(r'paac::', '', n)
    m = re.search(r'dmm_kernel<Dmm<Geom<([0-9, ]+)>, (true|false), (\d), (\d), (\d), (\d), (\d), (\d), (\d), (\d+), (\d), (true|false), (\d)(?:, (\d))?> ?>', n)
    if m:
        return 'dmm<G%s u8=%s ap%s bp%s T%sx%s W%s,%s,%s epi%s bias=%s pf%s%s>' % (
            m.group(1).replace(' ', ''), m.group(2)[0], m.group(3), m.group(4), m.group(5), m.group(6), m.group(7), m.group(8),
            m.group(9), m.group(11), m.group(12)[0], m.group(13),
            {None: "", "0": "", "1": " bf16-exact", "2": " bf16-split"}[m.group(14)])
    return n.split('(')[0][:70]

# 1. rocprofv3 --kernel-trace --stats summary (verbatim csv) + one-cycle timeline
stats = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, "profiles/%s_rocprofv3_kernel_stats.csv" % tag)
trace = glob.glob(os.path.join(src, "stats", "*", "*_kernel_trace.csv"))[0]
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'rmsprop_kernel' in r['Kernel_Name']]
k0 = len(marks) // 3         # steady-state part of the hipGraph phase (the tail of the trace is the eager, event-bracketed pass)
# the profiler occasionally stalls the queue for milliseconds while it drains its buffers: take the shortest of a few
# consecutive cycles
def wall(k):
    return int(rows[marks[k + 1] + 1]['Start_Timestamp']) - int(rows[marks[k] + 1]['Start_Timestamp'])
k = min(range(k0, min(k0 + 8, len(marks) - 2)), key=wall)
i0, i1 = marks[k] + 1, marks[k + 1] + 1
t0 = int(rows[i0]['Start_Timestamp'])
with open("profiles/%s_cycle_timeline.txt" % tag, "w") as f:
    f.write("# one steady-state PAAC cycle (bench.py under rocprofv3 --kernel-trace): start_us end_us dur_us workgroups x threads kernel\n")
    for r in rows[i0:i1]:
        s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
        wg = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // int(r['Workgroup_Size_X'])
        f.write("%8.1f %8.1f %6.1f  %6d x %3d  %s\n" % (s / 1000, e / 1000, (e - s) / 1000, wg, int(r['Workgroup_Size_X']), short(r['Kernel_Name'])))
    f.write("# cycle wall %.1f us, %d kernels\n" % ((int(rows[i1]['Start_Timestamp']) - t0) / 1000, i1 - i0))

# 2. PMC traffic per kernel (separate passes; FETCH_SIZE doubled per MI355X_MICROARCH.md: gfx950 tallies 128-B requests at 64 B)
def pmc(name, counter):
    f = glob.glob(os.path.join(src, name, "*", "*counter_collection.csv"))[0]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        k = (short(r['Kernel_Name']), r.get('Grid_Size', ''))
        agg[k][0] += float(r['Counter_Value']); agg[k][1] += 1
    return agg
fe, wr = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
traffic = {}
with open("profiles/%s_pmc_traffic.csv" % tag, "w") as f:
    f.write("kernel,grid_size,launches,FETCH_SIZE_KB_raw_per_launch,fetch_bytes_corrected_x2,WRITE_SIZE_KB_per_launch,hbm_bytes_per_launch\n")
    for k in sorted(fe, key=lambda k: -fe[k][0]):
        fk = fe[k][0] / fe[k][1]
        wk = wr.get(k, [0, 1])[0] / max(1, wr.get(k, [0, 1])[1])
        tot = (2 * fk + wk) * 1024
        traffic["%s|%s" % k] = tot
        f.write("\"%s\",%s,%d,%.1f,%.0f,%.1f,%.0f\n" % (k[0], k[1], fe[k][1], fk, 2 * fk * 1024, wk, tot))
json.dump(traffic, open("profiles/%s_pmc_traffic.json" % tag, "w"), indent=1)

# 2b. MFMA utilisation per kernel (its own pass): SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy cycles summed over the
# SIMDs; GRBM_GUI_ACTIVE is the kernel's clock cycles summed over the 8 XCDs -> share = busy / (GUI / 8 * 256 CUs * 4 SIMDs)
try:
    f = glob.glob(os.path.join(src, "pmc_mfma", "*", "*counter_collection.csv"))[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(f)):
        k = (short(r['Kernel_Name']), r.get('Grid_Size', ''))
        agg[k][r['Counter_Name']][0] += float(r['Counter_Value'])
        agg[k][r['Counter_Name']][1] += 1
    with open("profiles/%s_pmc_mfma.csv" % tag, "w") as out:
        # average duration of the same kernel in the kernel-trace pass (un-countered): the counter pass serialises the
        # dispatches and GRBM_GUI_ACTIVE of a 10 us kernel includes its drain, so the share against wall time x 2.4 GHz is
        # given beside the derived-counter formula MfmaUtil = busy / (GUI_ACTIVE * SIMDs)
        dur = collections.defaultdict(lambda: [0.0, 0])
        for r in rows:
            wgs = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
            d = dur[(short(r['Kernel_Name']), str(wgs))]
            d[0] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
            d[1] += 1
        out.write("kernel,grid_size,launches,SQ_VALU_MFMA_BUSY_CYCLES_per_launch,GRBM_GUI_ACTIVE_per_launch,SQ_BUSY_CYCLES_per_launch,"
                  "mfma_busy_share_vs_gui_active,avg_duration_us_kernel_trace_pass,mfma_busy_share_vs_duration_at_2.4GHz\n")
        def per(k, c):
            v = agg[k].get(c)
            return v[0] / v[1] if v and v[1] else 0.0
        for k in sorted(agg, key=lambda k: -per(k, "SQ_VALU_MFMA_BUSY_CYCLES") * agg[k]["SQ_VALU_MFMA_BUSY_CYCLES"][1]):
            busy, gui, sq = per(k, "SQ_VALU_MFMA_BUSY_CYCLES"), per(k, "GRBM_GUI_ACTIVE"), per(k, "SQ_BUSY_CYCLES")
            share = busy / (gui / 8.0 * 256 * 4) if gui > 0 else 0.0
            dk = dur.get(k)
            us = dk[0] / dk[1] / 1000.0 if dk and dk[1] else 0.0
            share2 = busy / 1024.0 / (us * 2400.0) if us > 0 else 0.0
            out.write("\"%s\",%s,%d,%.0f,%.0f,%.0f,%.4f,%.2f,%.4f\n" % (k[0], k[1], agg[k]["GRBM_GUI_ACTIVE"][1], busy, gui, sq, share,
                                                                  us, share2))
except (IndexError, KeyError) as exc:
    print("no MFMA pass:", exc)

# per kernel-family table in bench.py's naming ("family[batch=B]"): dmm families are read off the template arguments
# (geometry + fragment pattern; the smaller launch of a family is the acting batch), the fused kernels off their names
PATTERNS = [("conv1_wgrad", "G84,84,4,20,20", "ap1"), ("conv2_wgrad", "G20,20,", "ap1"),
            ("conv3_wgrad", "G9,9,64,7,7,1,0,0", "ap1"),
            ("fc_fwd", "G1,1,3136", "ap0"), ("fc_wgrad", "G1,1,3136", "ap1"), ("fc_dgrad", "G1,1,512", "ap0"),
            ("conv1_fwd", "G84,84,4,20,20", "ap0"), ("conv2_fwd", "G20,20,", "ap0"), ("conv3_fwd", "G9,9,64,7,7,1,0,0", "ap0"),
            ("conv3_dgrad", "G7,7,64,9,9", "ap0"), ("conv2_dgrad", "G9,9,64,10,10", "ap0")]
meta = json.loads(open(os.path.join(src, "bench_default.json")).read().strip().splitlines()[-1])
n_act, n_train = meta["config"]["envs_per_gpu"], meta["config"]["envs_per_gpu"] * meta["config"]["t_max"]
n_fwd = n_train + n_act       # the training forward carries the bootstrap rows
fam = {}
for name, gpat, ap in PATTERNS:
    ks = sorted([k for k in fe if gpat in k[0] and (" %s " % ap) in k[0]], key=lambda k: traffic["%s|%s" % k])
    if not ks:
        continue
    batches = [n_act, n_fwd] if len(ks) == 2 else ([n_train] if "grad" in name else [n_fwd if name.endswith("fwd") else n_act])
    for k, b in zip(ks, batches):
        fam["%s[batch=%d]" % (name, b)] = traffic["%s|%s" % k]
FUSED = [("conv_tower", "tower_kernel<TowerGeom", None), ("fc_fwd", "fc_heads_kernel", n_act),
         ("sample_env_step", "synth_step_a_mth_kernel", n_act), ("sample_env_step", "synth_step_a_mt_kernel", n_act),
         ("dgrad_tower", "dgrad_tower_kernel", n_train), ("heads_fwd", "heads_fwd_kernel", n_fwd),
         ("heads_bwd", "heads_bwd_kernel", n_train), ("heads_bwd", "heads_train_kernel", n_train),
         ("fc_conv3_wgrad", "dmm_pair_kernel<Dmm<Geom<1, 1, 3136", n_train),
         ("conv2_conv1_wgrad", "dmm_pair_kernel<Dmm<Geom<20, 20, 32", n_train),
         ("grad_finalize", "grad_finalize_kernel", n_train),
         ("nstep_returns", "nstep_returns_kernel", n_train)]
for name, pat, b in FUSED:
    ks = sorted([k for k in fe if pat in k[0]], key=lambda k: traffic["%s|%s" % k])
    if name == "conv_tower":           # acting regions variant (smaller traffic) and whole-sample training variant
        for k, bb in zip(ks, [n_act, n_fwd]):
            fam["conv_tower[batch=%d]" % bb] = traffic["%s|%s" % k]
    elif ks:
        fam["%s[batch=%d]" % (name, b)] = traffic["%s|%s" % ks[-1]]
rk = [k for k in fe if "rmsprop_kernel" in k[0] or "sumsq_kernel" in k[0] or "norm_kernel" in k[0]]
if rk:
    fam["clip_rmsprop[batch=%d]" % 0] = sum(traffic["%s|%s" % k] for k in rk)
json.dump({"workload": meta["config"]["workload"], "bytes_per_launch": fam,
           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over bench.py; (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                     "(gfx950 FETCH_SIZE tallies 128-B requests at 64 B); average per launch"},
          open("profiles/%s_traffic_by_family.json" % tag, "w"), indent=1)

# 2c. BASELINE configs[2]'s raw-frame path (256 environments, raw 210x160 screen pairs -> preprocess kernel): kernel stats of
# its own rocprofv3 run, PMC traffic of its own passes, and the per-family table bench.py looks up for that workload
try:
    rstats = glob.glob(os.path.join(src, "stats_raw256", "*", "*_kernel_stats.csv"))[0]
    shutil.copy(rstats, "profiles/%s_raw256_rocprofv3_kernel_stats.csv" % tag)

    def pmc_dir(name, counter):
        f = glob.glob(os.path.join(src, name, "*", "*counter_collection.csv"))[0]
        agg = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            k = (short(r['Kernel_Name']), r.get('Grid_Size', ''))
            agg[k][0] += float(r['Counter_Value']); agg[k][1] += 1
        return agg
    rfe, rwr = pmc_dir("pmc_fetch_raw256", "FETCH_SIZE"), pmc_dir("pmc_write_raw256", "WRITE_SIZE")
    rtraffic = {}
    with open("profiles/%s_raw256_pmc_traffic.csv" % tag, "w") as f:
        f.write("kernel,grid_size,launches,FETCH_SIZE_KB_raw_per_launch,fetch_bytes_corrected_x2,WRITE_SIZE_KB_per_launch,hbm_bytes_per_launch\n")
        for k in sorted(rfe, key=lambda k: -rfe[k][0]):
            fk = rfe[k][0] / rfe[k][1]
            wk = rwr.get(k, [0, 1])[0] / max(1, rwr.get(k, [0, 1])[1])
            rtraffic[k] = (2 * fk + wk) * 1024
            f.write("\"%s\",%s,%d,%.1f,%.0f,%.1f,%.0f\n" % (k[0], k[1], rfe[k][1], fk, 2 * fk * 1024, wk, rtraffic[k]))
    rmeta = json.loads(open(os.path.join(src, "bench_256envs_raw.json")).read().strip().splitlines()[-1])
    rn = rmeta["config"]["envs_per_gpu"]
    rfam = {}
    for name, pat in (("preprocess_stack", "preprocess_stack_kernel"), ("sample_env_step", "synth_step_a_mt_kernel"),
                      ("env_step", "synth_raw_kernel"), ("fc_fwd", "fc_heads_kernel"), ("heads_fwd", "heads_finish_rows_kernel")):
        ks = [k for k in rtraffic if pat in k[0]]
        if ks:
            rfam["%s[batch=%d]" % (name, rn)] = max(rtraffic[k] for k in ks)
    ks = sorted([k for k in rtraffic if "tower_kernel<TowerGeom" in k[0]], key=lambda k: rtraffic[k])
    for k, bb in zip(ks, [rn, rn * (rmeta["config"]["t_max"] + 1)]):
        rfam["conv_tower[batch=%d]" % bb] = rtraffic[k]
    json.dump({"workload": rmeta["config"]["workload"], "bytes_per_launch": rfam,
               "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over bench.py --envs 256 --raw-frames; "
                         "(2*FETCH_SIZE + WRITE_SIZE)*1024; average per launch"},
              open("profiles/%s_raw256_traffic_by_family.json" % tag, "w"), indent=1)
except (IndexError, KeyError, FileNotFoundError) as exc:
    print("no raw-frame passes:", exc)

# 3. bench lines + tuner log
for name in sorted(os.path.basename(f) for f in glob.glob(os.path.join(src, "bench_*.json"))) + ["tune_gemm.txt"]:
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, "profiles/%s_%s" % (tag, name))
print(open("profiles/%s_cycle_timeline.txt" % tag).read()[-600:])
