cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_hci/trace -- python3 tools/bench_hci.py > gpurun_out/prof_hci.log 2>&1
tail -2 gpurun_out/prof_hci.log | cut -c1-600
python3 - <<'PY'
import csv,glob
best=None
for f in glob.glob('gpurun_out/prof_hci/trace/*/*kernel_stats.csv'):
    rows=list(csv.DictReader(open(f)))
    if best is None or len(rows)>len(best): best=rows
out=open('gpurun_out/r02_hci_1e-4_rocprof_summary.txt','w')
out.write("# rocprofv3 --kernel-trace --stats -- python3 tools/bench_hci.py   (C2 cc-pVDZ, eps_var 1e-4, eps_pt 1e-6: variational stage, PT2, final sparse-H build + matvec)\n")
out.write("%-66s %8s %12s %10s\n"%("kernel","calls","avg_us","percent"))
for r in sorted(best,key=lambda r:-float(r['Percentage']))[:28]:
    out.write("%-66s %8s %12.2f %10s\n"%(r['Name'][:64], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
out.close()
print(open('gpurun_out/r02_hci_1e-4_rocprof_summary.txt').read())
PY
