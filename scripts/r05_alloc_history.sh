# Which kernels of the store-fed step are slower when a 28x28 side object ran earlier in the same process?  Two kernel
# traces of bench.py whose LAST timed loop is the resident-store ingest step: (a) alone, (b) behind the in-process 28x28 side.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; python -c "import torch" >/dev/null 2>&1
for tag in alone behind; do
  if [ $tag = alone ]; then S="joint_training_ingest"; else S="joint_training_ingest,joint_training_28x28"; fi
  rm -rf /tmp/prof_ah_$tag
  timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_ah_$tag -o ah -- python bench.py --no-cpu-baseline --no-roofline --sides $S --config5-inprocess > gpurun_out/r05_ah_$tag.log 2>&1
  python profiles/summarize.py --steady 15 $(find /tmp/prof_ah_$tag -name '*_results.db' | head -1) > gpurun_out/r05_ah_${tag}_steady.txt 2>&1
  head -3 gpurun_out/r05_ah_${tag}_steady.txt | cut -c1-200
done
python - <<'PY'
def load(p):
    d={}
    for l in open(p):
        if l.startswith('#') or l.startswith('kernel'): continue
        f=l.split()
        if len(f)>=5: d[f[0]]=(float(f[1]),float(f[2]))
    return d
a,b=load('gpurun_out/r05_ah_alone_steady.txt'),load('gpurun_out/r05_ah_behind_steady.txt')
rows=sorted(((b.get(k,(0,0))[1]-a.get(k,(0,0))[1],k) for k in set(a)|set(b)),reverse=True)
print("largest per-iteration differences (behind - alone), ms:")
for d,k in rows[:12]: print("%8.3f  %6.1f/%6.1f calls  %8.3f -> %8.3f  %s"%(d,a.get(k,(0,0))[0],b.get(k,(0,0))[0],a.get(k,(0,0))[1],b.get(k,(0,0))[1],k[:90]))
PY
