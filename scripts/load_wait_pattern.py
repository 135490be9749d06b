#!/usr/bin/env python3
# scripts/load_wait_pattern.py <object.o> [...] : per gfx950 kernel of the objects, the ORDER of its global loads (L), stores (S),
# s_waitcnt vmcnt(N) (wN) and barriers (|), and how often a load is followed by vmcnt(0) at once ("Lw0xN") -- a run of "Lw0Lw0Lw0..." is a
# chain of memory round trips the source did not mean (a load under a per-row guard in an unrolled loop).
import re,sys,subprocess,tempfile,os
B='/opt/rocm/lib/llvm/bin'
for o in sys.argv[1:]:
    t=tempfile.mkdtemp()
    subprocess.run([B+'/llvm-objcopy','--dump-section','.hip_fatbin=%s/fat.bin'%t,o],stderr=subprocess.DEVNULL)
    subprocess.run([B+'/clang-offload-bundler','--unbundle','--type=o','--input=%s/fat.bin'%t,'--targets=hipv4-amdgcn-amd-amdhsa--gfx950','--output=%s/dev.o'%t],stderr=subprocess.DEVNULL)
    out=subprocess.run([B+'/llvm-objdump','-d','%s/dev.o'%t],capture_output=True,text=True).stdout
    funcs={};cur=None
    for l in out.split('\n'):
        m=re.match(r'^[0-9a-f]+ <(.*)>:',l)
        if m: cur=m.group(1); funcs[cur]=[]; continue
        if cur: funcs[cur].append(l)
    for name,body in funcs.items():
        seq=[]
        for l in body:
            if re.search(r'\b(global_load|buffer_load|flat_load)',l): seq.append('L')
            elif re.search(r'\b(global_store|flat_store)',l): seq.append('S')
            elif 's_waitcnt' in l and 'vmcnt' in l:
                m=re.search(r'vmcnt\((\d+)\)',l); seq.append('w%s'%m.group(1))
            elif 's_barrier' in l: seq.append('|')
        sq=''.join(seq)
        n=len(re.findall(r'Lw0',sq))
        dem=name
        dem=re.sub(r'pclhip::\(anonymous namespace\)::','',dem)[:75]
        print('%-14s %-60s Lw0x%-3d %s'%(os.path.basename(o),dem,n,sq[:150]))
