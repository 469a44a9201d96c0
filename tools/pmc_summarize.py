"""Dev tool: per-launch medians of the PMC passes tools/pmc_collect.sh wrote under OUT (p*/**/*counter_collection.csv) -> OUT/summary.{txt,json}.
   python tools/pmc_summarize.py OUT POINTS_PER_LAUNCH "what was profiled"      (also usable here on the CSVs a gpurun call merged back)"""
import collections, csv, glob, hashlib, json, os, sys
out, pts, what = sys.argv[1], int(sys.argv[2]), sys.argv[3]
root = os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, 'p*/**/*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if 'fused' in r['Kernel_Name'] and int(r['Grid_Size']) >= 256 * 512:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
js = {}
with open(os.path.join(out, 'summary.txt'), 'w') as o:
    for k in sorted(acc):
        v = sorted(acc[k])[len(acc[k]) // 2:]          # the full-size launches are the largest values: median of the top half
        js[k + '_median_per_launch'] = sorted(v)[len(v) // 2]
        line = f'{k:32s} {sorted(v)[len(v) // 2]:.4e}  (n={len(acc[k])})'
        print(line)
        o.write(line + '\n')
if 'FETCH_SIZE_median_per_launch' in js and 'WRITE_SIZE_median_per_launch' in js:
    # FETCH_SIZE / WRITE_SIZE count KB; gfx950: streamed 16-byte reads are under-counted by 2 (MI355X_MICROARCH.md)
    js['hbm_bytes_per_launch'] = 1024.0 * (2.0 * js['FETCH_SIZE_median_per_launch'] + js['WRITE_SIZE_median_per_launch'])
g = lambda k: js.get(k + '_median_per_launch')
if g('TCC_EA0_RDREQ_sum') is not None and g('TCC_EA0_WRREQ_sum') is not None:
    # exact request sizes on the L2 <-> fabric interface: 32-byte and 64-byte requests counted separately
    js['ea_read_bytes_per_launch'] = 32.0 * g('TCC_EA0_RDREQ_32B_sum') + 64.0 * (g('TCC_EA0_RDREQ_sum') - g('TCC_EA0_RDREQ_32B_sum'))
    js['ea_write_bytes_per_launch'] = 64.0 * g('TCC_EA0_WRREQ_64B_sum') + 32.0 * (g('TCC_EA0_WRREQ_sum') - g('TCC_EA0_WRREQ_64B_sum'))
h = hashlib.sha256()
for f in ('pinn_fused.hpp', 'pinn_device.hpp', 'pinn_host.hpp'):
    h.update(open(os.path.join(root, 'pinn_elastodynamics_amd/csrc', f), 'rb').read())
js['points_per_launch'] = pts
js['kernel_source_sha'] = h.hexdigest()[:16]      # bench.py refuses to quote these bytes for other kernel sources
js['note'] = (what + '; tools/pmc_collect.sh: one rocprofv3 --pmc pass per group of <= 4 counters, --kernel-trace only. FETCH_SIZE/WRITE_SIZE are in KB; '
              'hbm_bytes = 2*FETCH (gfx950 correction) + WRITE. These L2<->fabric counters include Infinity-Cache hits.')
json.dump(js, open(os.path.join(out, 'summary.json'), 'w'), indent=1, sort_keys=True)
