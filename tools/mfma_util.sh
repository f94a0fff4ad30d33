# MFMA-busy of the GEMM kernels in isolation: rocprofv3 --pmc MfmaUtil over tools/bench_gemm.py (run on the GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mfu && timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES -d /tmp/mfu -o m --output-format csv -- python3 $ROOT/tools/bench_gemm.py ${1:-15872} > /tmp/mfu.log 2>&1
python3 - <<'PY'
import csv, collections, glob
f = glob.glob('/tmp/mfu/**/m_counter_collection.csv', recursive=True)[0]
t = glob.glob('/tmp/mfu/**/m_kernel_trace.csv', recursive=True)[0]
dur = {}
with open(t) as fh:
    for r in csv.DictReader(fh):
        dur[r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
acc = collections.defaultdict(lambda: {'n': 0, 'busy': 0.0, 'gui': 0.0, 'ns': 0.0})
seen = set()
with open(f) as fh:
    for r in csv.DictReader(fh):
        name = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'k_gemm' not in name: continue
        key = (name, int(r['Grid_Size']) // max(1, int(r['Workgroup_Size'])))
        a = acc[key]
        if r['Counter_Name'] == 'SQ_VALU_MFMA_BUSY_CYCLES': a['busy'] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE': a['gui'] += float(r['Counter_Value'])
        if r['Dispatch_Id'] not in seen:
            seen.add(r['Dispatch_Id']); a['n'] += 1; a['ns'] += dur.get(r['Dispatch_Id'], 0)
print('rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES -- python3 tools/bench_gemm.py <M ...>')
print('SQ_VALU_MFMA_BUSY_CYCLES per launch = (number of v_mfma_f32_32x32x2_f32) x 64 exactly: the matrix-pipe cycles the launch needs.')
print('GRBM_GUI_ACTIVE is summed over the 8 XCDs: gui/8 = shader cycles the launch was active; gui/8/duration = the clock it ran at.')
print('MFMA busy = busy / (1024 SIMDs x gui/8).  (durations here are under counter collection: a few % longer than unprofiled)')
print('kernel | workgroups | launches | avg us | clock GHz | MFMA busy | TFLOP/s at that duration')
for key, c in sorted(acc.items()):
    if c['busy'] and c['gui'] and c['n']:
        n = c['n']
        cyc = c['gui'] / n / 8
        us = c['ns'] / n / 1e3
        flops = c['busy'] / n / 64 * 4096
        print(f"{key[0]} | {key[1]} | {n} | {us:.1f} | {cyc / (us * 1e3):.2f} | {c['busy'] / n / (1024 * cyc):.3f} | {flops / (us * 1e-6) / 1e12:.1f}")
PY
