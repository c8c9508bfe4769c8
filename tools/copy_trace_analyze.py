import csv, glob, statistics as st, sys
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 250
f = glob.glob('/tmp/ct/**/*memory_copy_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
h2d = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows
             if 'HOST_TO_DEVICE' in r['Direction'] and int(r['End_Timestamp']) - int(r['Start_Timestamp']) > 100000)
print(len(h2d), 'large H2D copies')
t0 = h2d[0][0]
base = st.median([(e - s) / 1e3 for s, e in h2d[:20]])
for c in range(0, len(h2d), batch):
    part = h2d[c:c + batch]
    if not part:
        break
    durs = [(e - s) / 1e3 for s, e in part]
    gaps = [(part[i + 1][0] - part[i][1]) / 1e3 for i in range(len(part) - 1)]
    slow = [i for i, d in enumerate(durs) if d > 1.5 * base]
    print('call %d: %.1f ms from first copy to last; copy us median %.0f; gaps sum %.1f ms; first slow copy: %s (%.1f ms, %.2f GB after the call began); slow copies %d of %d' % (
        c // batch, (part[-1][1] - part[0][0]) / 1e6, st.median(durs), sum(gaps) / 1e3,
        slow[0] if slow else None, (part[slow[0]][0] - part[0][0]) / 1e6 if slow else 0,
        sum(d for d in durs[:slow[0]]) * 56e-6 if slow else 0, len(slow), len(part)))
k = glob.glob('/tmp/ct/**/*kernel_trace.csv', recursive=True)
kr = list(csv.DictReader(open(k[0])))
for c in range(0, len(h2d), batch):
    part = h2d[c:c + batch]
    if not part:
        break
    lo, hi = part[0][0], part[-1][1]
    blits = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in kr
             if 'copyBuffer' in r['Kernel_Name'] and int(r['End_Timestamp']) > lo and int(r['Start_Timestamp']) < hi]
    print('call %d: %d __amd_rocclr_copyBuffer kernels in the window, durations us: median %.1f max %.1f; queues of all kernels: %s' % (
        c // batch, len(blits), st.median(blits) if blits else 0, max(blits) if blits else 0,
        sorted(set(r['Queue_Id'] for r in kr if int(r['End_Timestamp']) > lo and int(r['Start_Timestamp']) < hi))))
import collections
for c in range(0, min(len(h2d), 2 * batch), batch):
    part = h2d[c:c + batch]
    lo, hi = part[0][0], part[-1][1]
    byq = collections.Counter((r['Queue_Id'], r['Stream_Id'], r['Kernel_Name'][:48]) for r in kr if int(r['End_Timestamp']) > lo and int(r['Start_Timestamp']) < hi)
    print('call %d kernels by (queue, stream, name):' % (c // batch))
    for key, n in sorted(byq.items()):
        print('   ', key, n)
cp = collections.Counter((r['Direction'], r['Stream_Id']) for r in rows)
print('copies by (direction, stream):', dict(cp))
