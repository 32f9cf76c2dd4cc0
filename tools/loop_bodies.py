#!/usr/bin/env python3
"""Instructions per iteration of every loop (backward branch) of one compiled kernel, by class -- arithmetic VALU, v_rsq / v_rcp, v_mov,
v_accvgpr_* (register traffic through the AGPRs), VMEM, DS, scalar:   python tools/loop_bodies.py <object or .so> <mangled-name substring>
(round 6: the per-marker cost of the reprojection fold IS its instruction count; EXPERIMENTS -1.7)"""
import re,sys,subprocess,os,tempfile,collections
LLVM="/opt/rocm/lib/llvm/bin"
path,filt=sys.argv[1],sys.argv[2]
td=tempfile.mkdtemp()
r=subprocess.run([f"{LLVM}/clang-offload-bundler","--list","--type=o",f"--input={path}"],capture_output=True,text=True)
if not [t for t in r.stdout.split() if "gfx950" in t]:
    fb=os.path.join(td,"fb")
    subprocess.run([f"{LLVM}/llvm-objcopy","-O","binary","--only-section=.hip_fatbin",path,fb],check=True)
    path=fb
    r=subprocess.run([f"{LLVM}/clang-offload-bundler","--list","--type=o",f"--input={path}"],capture_output=True,text=True)
tgt=[t for t in r.stdout.split() if "gfx950" in t][0]
co=os.path.join(td,"co")
subprocess.run([f"{LLVM}/clang-offload-bundler","--unbundle","--type=o",f"--input={path}",f"--targets={tgt}",f"--output={co}"],check=True)
dis=subprocess.run([f"{LLVM}/llvm-objdump","-d",co],capture_output=True,text=True).stdout
cur=None; ins=[]
for line in dis.splitlines():
    m=re.match(r"^([0-9a-f]+) <(.+)>:$",line)
    if m:
        if cur and filt in cur: break
        cur=m.group(2); ins=[]; continue
    if cur and filt in cur:
        m=re.match(r"^\s+(\S+)\s+(.*?)//\s*([0-9A-F]+):",line)
        if m: ins.append((int(m.group(3),16), m.group(1), m.group(2)))
print(cur, len(ins))
addr={a:i for i,(a,_,_) in enumerate(ins)}
for i,(a,op,args) in enumerate(ins):
    if op.startswith("s_cbranch") or op=="s_branch":
        m=re.search(r"<.*\+0x([0-9a-f]+)>",line) 
# branch targets: objdump prints as "s_cbranch_scc1 65532" relative words; compute target
for i,(a,op,args) in enumerate(ins):
    if op.startswith("s_cbranch") or op=="s_branch":
        off=int(args.split()[0])
        if off>=32768: off-=65536
        t=a+4+off*4
        if t<a and t in addr:
            body=ins[addr[t]:i+1]
            c=collections.Counter()
            for _,o,_ in body:
                if o.startswith("v_accvgpr"): c["accvgpr"]+=1
                elif o.startswith("v_mov"): c["v_mov"]+=1
                elif o.startswith(("v_rsq","v_rcp")): c["trans"]+=1
                elif o.startswith("v_"): c["valu"]+=1
                elif o.startswith(("buffer_","global_")): c["vmem"]+=1
                elif o.startswith("ds_"): c["ds"]+=1
                elif o.startswith("s_waitcnt"): c["waitcnt"]+=1
                elif o.startswith("s_"): c["salu"]+=1
            print(f"loop {t:#x}..{a:#x}: {len(body)} instr", dict(c))
