#!/bin/bash
# dev tool (round 5): the 2^20 BLS12-381 transform (bench.py's ntt leg) with / without the LDS-staged twiddles, alternating
R=$PWD; O=$R/gpurun_out; mkdir -p $O
for rep in 1 2 3; do for v in ${@:-0 1}; do
  echo "rep $rep MANTA_NTT_TWL=$v $(MANTA_NTT_TWL=$v python -c "
import bench, json
o = bench.ntt_bench(None)
print(' '.join('%s %.4f (pass %.1f us)' % (k, o[k]['device_ms'], o[k]['us_per_pass']) for k in ('fft','ifft','coset_fft','coset_ifft')))
" 2>&1 | tail -1)"
done; done | tee $O/ntt_ab_r5.txt
