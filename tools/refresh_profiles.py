#!/usr/bin/env python
"""Dev tool: write profiles/r01_{kernel_stats.md,bench_line.json} and r01c_pmc_traffic.json from the gpurun_out/ results of
the end-of-round measurement command (see the header this script writes).  Usage: refresh_profiles.py <stats_dir> <log> <fetch_db> <write_db>"""
import json, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stats_db, log, fdb, wdb = sys.argv[1:5]
d = json.loads([l for l in open(log) if l.startswith('{')][0])
b = json.load(open(os.path.join(ROOT, 'gpurun_out', 'r01_bench_line.json')))
json.dump(b, open(os.path.join(ROOT, 'profiles', 'r01_bench_line.json'), 'w'))
table = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'rocpd_summary.py'), stats_db], capture_output=True, text=True).stdout
kavg = [l for l in table.splitlines() if 'k_transe_pair_sampled<32, 4, 4, false>' in l][0].split('|')[4].strip()
hdr = f"""# rocprofv3 --kernel-trace --stats, round 1, final build

Command (MI355X box): `rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline`
(ROCm 7.2 rocprofv3 writes a rocpd SQLite database; this table is `tools/rocpd_summary.py` over it = the kernel_stats view.)

bench.py line of the same (profiled) run: value {d['value']/1e6:.1f} M scored triples/s, ms_per_step {d['ms_per_step']:.4f}, roofline.avg_launch_ms {d['roofline']['avg_launch_ms']:.4f} (HIP events around each launch: they include the dispatch latency in front of the kernel, which the profiler inflates; the table's {kavg} µs avg for `k_transe_pair_sampled<32, 4, 4, false>` is the pure kernel time of the same launches), eval {d['eval']['value']/1e6:.2f} M test triples/s
Un-profiled `python bench.py` of the same build on the same box (profiles/r01_bench_line.json): {b['value']/1e6:.1f} M scored triples/s, {b['ms_per_step']:.4f} ms/step, kernel {b['roofline']['avg_launch_ms']:.4f} ms (frac {b['roofline']['frac']:.3f}), eval {b['eval']['value']/1e6:.2f} M test triples/s, default-batch (B=128) leg {b['train_reference_default_batch']['ms_per_step']*1e3:.1f} µs/step, CPU port {b['cpu_baseline']['value']/1e6:.1f} M/s on {b['cpu_baseline']['cores']} threads

Rows: `k_transe_pair_sampled<32, 4, 4, false>` = the timed train step's fused kernel (B=32768: sampler + both scores + hinge + backward);
`k_opt<1, true>` = dense Adam (+ grad zeroing; in the B=128 leg also the next step's device-resident state); `k_transe_pair_sampled<32, 4, 1, false>` = the
B=128 default-batch leg (hipGraph replay, two launches per step); `k_eval_*` = the filtered-rank pipeline (4 passes over 8192 test triples).
PMC traffic of the same command: r01c_pmc_traffic.json.

"""
open(os.path.join(ROOT, 'profiles', 'r01_kernel_stats.md'), 'w').write(hdr + table)
tmp = os.path.join(ROOT, 'gpurun_out', '_pmc_tmp.json')
subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'rocpd_pmc.py'), tmp, 'x', fdb, wdb], check=True)
old = json.load(open(os.path.join(ROOT, 'profiles', 'r01c_pmc_traffic.json')))
old['kernels'] = json.load(open(tmp))['kernels']
json.dump(old, open(os.path.join(ROOT, 'profiles', 'r01c_pmc_traffic.json'), 'w'), indent=1)
print('refreshed: value %.1f M/s, kernel %.2f us (rocprof %s us), eval %.2f M/s' % (b['value']/1e6, b['roofline']['avg_launch_ms']*1e3, kavg, b['eval']['value']/1e6))
