cd /root/repo; export TMPDIR=/tmp
V=$PWD/cfmm-routing-code_amd/cfmm/variants
for rep in 1 2; do for spec in default kwarm kwarm:CFMM_KWARM=0; do
 lib=${spec%%:*}; envs=""; [ "$spec" != "$lib" ] && envs=${spec#*:}
 L=$V/libcfmm_hip_$lib.so; [ "$lib" = default ] && L=
 env $envs CFMM_LIB=$L python tools/microbench.py --config C3 --tag $spec --solves 5 --reps 300 --buckets 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print(r['tag'], 'all %.2f'%r['eval_all_us'], {k.replace('eval_kernel[','').replace(' only]',''):v for k,v in r['buckets'].items()})"
done; done
