cd /tmp && export TMPDIR=/tmp
for n in 768 1536; do timeout 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05_en_trace_$n -o t -- python $GRAFT_REPO_ROOT/tools/chain_trace.py $n > /dev/null 2>&1; done
cd $GRAFT_REPO_ROOT
for n in 768 1536; do echo "== $n"; python tools/chain_trace_report.py gpurun_out/r05_en_trace_$n 24; done
