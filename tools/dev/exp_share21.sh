cd $GRAFT_REPO_ROOT
LPSLAM_HIP_SHARE_TRACE=1 LPSLAM_DEV_FLAT=1 timeout -k 10 200 python tools/dev_tracker_multi.py 8 120 > gpurun_out/tr21.log 2>&1
python3 - <<'PY'
import re, collections
req = collections.defaultdict(list)
for l in open('gpurun_out/tr21.log', errors='ignore'):
    m = re.match(r'req (.+?): to launch (\d+) us, launch to done (\d+) us', l)
    if m: req[m.group(1)].append((int(m.group(2)), int(m.group(3))))
import statistics as st
for k, v in req.items():
    v = v[len(v)//4:]          # steady part
    a = [x[0] for x in v]; b = [x[1] for x in v]
    q = lambda x, p: sorted(x)[int(p * (len(x) - 1))]
    print("%-28s n %5d   to launch mean %4.0f p50 %4d p90 %4d   launch->done mean %4.0f p50 %4d p90 %4d" % (k, len(v), st.mean(a), q(a, .5), q(a, .9), st.mean(b), q(b, .5), q(b, .9)))
PY
grep "managers:" gpurun_out/tr21.log
