#!/bin/bash
# Dev tool (GPU box): per-kernel PMC counters of an experiment library's fused launch, several passes of <= 4 counters.
#   tools/pmc_collect.sh NAME [WIDTH | nc3d | plate | plate70 | conf]     (library build/exp/NAME/libpinn_hip.so; output gpurun_out/pmc_NAME[_WIDTH]/)
# WIDTH (80, 100): the launches of tools/wide_time.py WIDTH NAME (1,000,000 points of the 8 x WIDTH net) instead of tools/exp_run.py (8x64, 2 M).
NAME=$1
WIDTH=$2
OUT=$PWD/gpurun_out/pmc_$NAME${WIDTH:+_$WIDTH}
if [ "$WIDTH" = "nc3d" ]; then CMD="$GRAFT_REPO_ROOT/tools/nc3d_time.py $NAME 6"; WHAT="fused 3-D kernel Fused<OpF16,3,128,10,5,false,4>, 1,000,000 points per launch (tools/nc3d_time.py)"
elif [ "$WIDTH" = "plate70" ]; then CMD="$GRAFT_REPO_ROOT/tools/plate_time.py 70 $NAME"; WHAT="plate collocation kernel of the reference's 8 x 70 net, Fused<OpF16,3,96,8,5>, 1,000,000 points per launch (tools/plate_time.py 70)"
elif [ "$WIDTH" = "conf" ]; then CMD="$GRAFT_REPO_ROOT/tools/conf_time.py"; WHAT="the reference's confined-domain net 6 x 140, Fused<OpF16,3,160,6,4>, 1,000,000 points per launch (tools/conf_time.py; the fused launches only)"
elif [ "$WIDTH" = "plate" ]; then CMD="$GRAFT_REPO_ROOT/tools/plate_time.py 64 $NAME"; WHAT="plate collocation kernel Fused<OpF16,3,64,8,5>, 1,000,000 points per launch (tools/plate_time.py)"
elif [ -n "$WIDTH" ]; then CMD="$GRAFT_REPO_ROOT/tools/wide_time.py $WIDTH $NAME"; WHAT="8x$WIDTH net, 1,000,000 points per launch (tools/wide_time.py)"
else CMD="$GRAFT_REPO_ROOT/tools/exp_run.py $NAME"; WHAT="fused_wave_kernel<OpF16,3,64,8,4>, 2,000,000 points per launch (tools/exp_run.py)"; fi
PTS=1000000; [ -z "$WIDTH" ] && PTS=2000000
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_IFETCH" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python $CMD > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('$OUT/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'fused' in r['Kernel_Name'] and int(r['Grid_Size']) >= 256 * 512:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
import json
js = {}
with open('$OUT/summary.txt', 'w') as o:
    for k in sorted(acc):
        v = acc[k]
        # the 2M-point launches are the largest values; take the median of the top half
        v = sorted(v)[len(v) // 2:]
        js[k + '_median_per_launch'] = sorted(v)[len(v)//2]
        line = f'{k:32s} {sorted(v)[len(v)//2]:.4e}  (n={len(acc[k])})'
        print(line); o.write(line + '\n')
if 'FETCH_SIZE_median_per_launch' in js and 'WRITE_SIZE_median_per_launch' in js:
    # FETCH_SIZE / WRITE_SIZE count KB; gfx950: streamed 16-byte reads are under-counted by 2 (MI355X_MICROARCH.md)
    js['hbm_bytes_per_launch'] = 1024.0 * (2.0 * js['FETCH_SIZE_median_per_launch'] + js['WRITE_SIZE_median_per_launch'])
g = lambda k: js.get(k + '_median_per_launch')
if g('TCC_EA0_RDREQ_sum') is not None and g('TCC_EA0_WRREQ_sum') is not None:
    # exact request sizes on the L2 <-> fabric interface: 32-byte and 64-byte requests counted separately
    js['ea_read_bytes_per_launch'] = 32.0 * g('TCC_EA0_RDREQ_32B_sum') + 64.0 * (g('TCC_EA0_RDREQ_sum') - g('TCC_EA0_RDREQ_32B_sum'))
    js['ea_write_bytes_per_launch'] = 64.0 * g('TCC_EA0_WRREQ_64B_sum') + 32.0 * (g('TCC_EA0_WRREQ_sum') - g('TCC_EA0_WRREQ_64B_sum'))
import hashlib
h = hashlib.sha256()
for f in ('pinn_fused.hpp', 'pinn_device.hpp', 'pinn_host.hpp'):
    h.update(open('$GRAFT_REPO_ROOT/pinn_elastodynamics_amd/csrc/' + f, 'rb').read())
js['points_per_launch'] = $PTS
js['kernel_source_sha'] = h.hexdigest()[:16]      # bench.py refuses to quote these bytes for other kernel sources
js['note'] = ('$WHAT; tools/pmc_collect.sh: one rocprofv3 --pmc pass per '
              'group of <= 4 counters, --kernel-trace only. FETCH_SIZE/WRITE_SIZE are in KB; hbm_bytes = 2*FETCH (gfx950 correction) + WRITE. These '
              'L2<->fabric counters include Infinity-Cache hits.')
json.dump(js, open('$OUT/summary.json', 'w'), indent=1, sort_keys=True)
PY
