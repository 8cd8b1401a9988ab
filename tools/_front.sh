# The rank pass (configs[3] as one of its 8 ranks) split into front / sampler / host: rocprofv3 kernel trace of tools/prof_rank.py ->
# per pass: GPU span of the front (first kernel .. first sampler launch), the time its kernels are busy, the sampler's span, and what
# the host's pass time has beyond the GPU span.  PB / PW as tools/prof_rank.py.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/front; rm -rf $O
PB=${PB:-64} PW=${PW:-8} PN=${PN:-3} timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/prof_rank.py > $O.log 2>&1
grep "rank pass" $O.log
python3 - <<'PY'
import csv,glob,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/front/*/*kernel_trace.csv')[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
S=lambda r:int(r['Start_Timestamp']); E=lambda r:int(r['End_Timestamp'])
samp=lambda r: 'residual_' in r['Kernel_Name'] or 'step_tail' in r['Kernel_Name']
# passes: maximal runs of sampler launches; the front of a pass = the kernels between the previous run's end and this run's start
runs=[]; cur=None
for i,r in enumerate(rows):
    if samp(r):
        if cur is None: cur=[i,i]
        cur[1]=i
    elif cur is not None and not samp(r):
        # a non-sampler kernel ends a run only if several follow (mel_finish etc.)
        runs.append(cur); cur=None
if cur: runs.append(cur)
runs=[r for r in runs if r[1]-r[0]>50]
prev_end=None
for k,(a,b) in enumerate(runs):
    j=a-1
    while j>=0 and not samp(rows[j]) and (prev_end is None or j>prev_end): j-=1
    fr=rows[j+1:a]
    # drop the epilogue of the previous pass (mel_finish and what follows within 50 us of the previous run)
    if prev_end is not None:
        fr=[r for r in fr if S(r)-E(rows[prev_end])>50_000 or 'mel_finish' not in r['Kernel_Name']]
    if fr:
        span=(S(rows[a])-S(fr[0]))/1e3; busy=sum(E(r)-S(r) for r in fr)/1e3
        print(f'pass {k}: front {len(fr)} kernels, span {span:.0f} us, busy {busy:.0f} us; sampler span {(E(rows[b])-S(rows[a]))/1e3:.0f} us; '
              f'first front kernel .. last sampler kernel {(E(rows[b])-S(fr[0]))/1e3:.0f} us')
        if k==len(runs)-1:
            top=sorted(fr,key=lambda r:S(r)-E(r))[:12]
            gaps=[(S(y)-E(x))/1e3 for x,y in zip(fr,fr[1:])]
            print('  gaps between front kernels: sum %.0f us, max %.1f us, median %.1f us'%(sum(gaps),max(gaps),sorted(gaps)[len(gaps)//2]))
            big=sorted(range(len(gaps)),key=lambda i:-gaps[i])[:6]
            for i in big: print('   gap %7.1f us between  %s  ->  %s'%(gaps[i], fr[i]['Kernel_Name'].replace('bsg::(anonymous namespace)::','')[:60], fr[i+1]['Kernel_Name'].replace('bsg::(anonymous namespace)::','')[:60]))
            for r in top: print('   %7.1f us  %s'%((E(r)-S(r))/1e3, r['Kernel_Name'].replace('bsg::(anonymous namespace)::','')[:90]))
    prev_end=b
PY
find $O -name "*.csv" -delete
