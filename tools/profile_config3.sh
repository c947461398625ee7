cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/pn -o n -- python /root/repo/tools/config3_bls_2_20.py 1 20 > /tmp/pn.txt 2>&1
tail -1 /tmp/pn.txt | cut -c1-200
python /root/repo/tools/rocprof_summary.py $(find /tmp/pn -name "*.db" | head -1) | head -40
