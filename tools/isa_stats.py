import re, collections, sys, subprocess, os, tempfile, shutil
obj=sys.argv[1]; frag=sys.argv[2]
tmp=tempfile.mkdtemp(); shutil.copy(obj, tmp+'/o.o')
subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objdump','--offloading','o.o'],cwd=tmp,capture_output=True)
co=[f for f in os.listdir(tmp) if 'gfx950' in f][0]
txt=subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objdump','-d','--no-show-raw-insn',tmp+'/'+co],capture_output=True,text=True).stdout
notes=subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf','--notes',tmp+'/'+co],capture_output=True,text=True).stdout
m=re.search(r"^[0-9a-f]+ <(_Z[^>]*"+frag+r"[^>]*)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", txt, re.S|re.M)
name=m.group(1)
lines=m.group(2).splitlines()
ins=[]
for l in lines:
    mm=re.match(r"\s*(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):(.*)$", l)
    if mm: ins.append((int(mm.group(3),16), mm.group(1), mm.group(2), mm.group(4)))
base=ins[0][0]; addr={a:i for i,(a,_,_,_) in enumerate(ins)}
br=[]
for i,(a,op,args,c) in enumerate(ins):
    if op.startswith('s_cbranch') or op=='s_branch':
        mm=re.search(r"\+0x([0-9a-fA-F]+)>", c)
        if mm and base+int(mm.group(1),16) in addr: br.append((i,addr[base+int(mm.group(1),16)],op))
bars=[i for i,x in enumerate(ins) if x[1]=='s_barrier']
loops=[(i-t,t,i) for i,t,op in br if t<=i and (not bars or any(t<=b<=i for b in bars))]
big=[l for l in loops if l[0]>=400]; _,lo,hi=min(big) if big else max(loops)      # the tick loop: the smallest loop around a barrier that is long enough to be it (the persistent-tile loop around it is larger)
fwd=[(t-i,i,t) for i,t,op in br if lo<=i<t<=hi and op=='s_cbranch_execz']
_,olo,ohi=max(fwd)
inner=[ins[i] for i in range(lo,hi+1) if not (olo<i<ohi)]
c=collections.Counter(x[1] for x in inner)
blk=re.search(r"\.name:\s+"+re.escape(name)+r".*?\n(.*?)\.symbol", notes, re.S)
vg=re.findall(r"\.vgpr_count:\s+(\d+)", notes); 
idx=[m.start() for m in re.finditer(re.escape(name), notes)]
seg=notes[max(0,idx[0]-1500):idx[0]+200]
print(name[:70])
print(' loop', hi-lo+1, 'outer', ohi-olo, 'inner', len(inner), '| inner: s_mov', c['s_mov_b32'], 'lane ops', sum(v for k,v in c.items() if 'lane' in k), 'valu', sum(v for k,v in c.items() if k.startswith('v_')), 'branches', sum(v for k,v in c.items() if 'branch' in k),
      '| scalar loads', sum(v for k,v in c.items() if k.startswith('s_load')), 'waits', c['s_waitcnt'])
print(' regs:', re.findall(r"\.(?:sgpr_count|sgpr_spill_count|vgpr_count|vgpr_spill_count|private_segment_fixed_size):\s+\d+", seg)[-5:])
shutil.rmtree(tmp)
