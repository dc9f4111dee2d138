"""development: summary of a rocprofv3 --kernel-trace CSV of a tracker_multi run (last `ms` milliseconds): per kernel launches, durations, queues, batch sizes"""
import csv, collections, re, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ms = float(sys.argv[2]) if len(sys.argv) > 2 else 230
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
    m = re.search(r'(k_[a-z_0-9]+|__amd_rocclr_\w+)', r['Kernel_Name']); r['n'] = m.group(1) if m else r['Kernel_Name'][:30]
tend = max(r['e'] for r in rows)
ph = [r for r in rows if r['s'] > tend - ms * 1e6]
names = collections.defaultdict(list)
for r in ph: names[r['n']].append(r)
print("%d kernels in the last %.0f ms" % (len(ph), ms))
for n, v in sorted(names.items(), key=lambda kv: -sum(r['e'] - r['s'] for r in kv[1])):
    d = [(r['e'] - r['s']) / 1e3 for r in v]
    wg = int(v[0]['Workgroup_Size_X']) * int(v[0]['Workgroup_Size_Y'])
    gy = collections.Counter(int(r['Grid_Size_Y']) // max(int(r['Workgroup_Size_Y']), 1) for r in v).most_common(4)
    gx = collections.Counter(int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1) for r in v).most_common(4)
    print('%-26s %5d launches, %8.1f us total, med %6.1f p90 %6.1f max %7.1f | blocks x %s y %s | queues %s' % (n, len(v), sum(d), statistics.median(d), sorted(d)[int(len(d) * 0.9)], max(d), gx, gy, dict(collections.Counter(r['Queue_Id'] for r in v))))
# busy fraction per queue
for q in sorted(set(r['Queue_Id'] for r in ph)):
    iv = sorted((r['s'], r['e']) for r in ph if r['Queue_Id'] == q)
    busy = 0; cur_s, cur_e = iv[0]
    for s, e in iv[1:]:
        if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print("queue %s: busy %.1f %% of the window, %d kernels" % (q, 100.0 * busy / (ms * 1e6), len(iv)))
# what runs on the queue of the shared matcher launches, and how long a launch waits behind its predecessor on that queue
pq = collections.Counter(r['Queue_Id'] for r in ph if r['n'] == 'k_proj_topk_req').most_common(1)
if pq:
    q = pq[0][0]
    on = sorted((r for r in ph if r['Queue_Id'] == q), key=lambda r: r['s'])
    print("queue %s (matchers): kernels by name:" % q, dict(collections.Counter(r['n'] for r in on).most_common(12)))
    tot = collections.defaultdict(float)
    for r in on: tot[r['n']] += (r['e'] - r['s']) / 1e3
    print("   time by name (us):", {k: round(v) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:10]})
