"""Bank-conflict count of the four LDS crossings of range_mixed.hip for a line length N = R1 * R2 * R3 and each pad 0..16:
    python tools/lds_layout_sim.py [N R1 R2 R3 T [f32]]  (default 13200 24 22 25 640; f32: one float plane at a time, MixCfg::PLANES)
prints LDS cycles (with conflicts, conflict-free) of the write and the read side of every crossing; pick the pads with the
smallest sums and put them into the MixCfg of that length."""
import itertools
import sys
N,R1,R2,R3,T=(int(x) for x in sys.argv[1:6]) if len(sys.argv) >= 6 else (13200,24,22,25,640)
F32=len(sys.argv) >= 7 and sys.argv[6] == "f32"      # 4-byte elements: the same bytes per clock, twice the lanes per group
def conflicts(addrs_per_lane, write):
    """addrs: list over lanes (64) of element address (8-byte elements, 4-byte with f32) or None (inactive). returns LDS cycles."""
    cyc=0
    if write:
        groups=[range(g*32,g*32+32) for g in range(2)] if F32 else [range(g*16,g*16+16) for g in range(4)]; nb=32
    else:
        groups=[range(0,64)] if F32 else [range(0,32),range(32,64)]; nb=64
    for g in groups:
        banks={}
        for l in g:
            a=addrs_per_lane[l]
            if a is None: continue
            for d in ((0,) if F32 else (0,1)):
                w=a if F32 else 2*a+d
                banks.setdefault(w%nb,set()).add(w)
        cyc+=max([len(v) for v in banks.values()],default=0)
    return cyc
def total(fn, nthreads, R, write):
    tot=0;ideal=0
    for w0 in range(0,T,64):
        for r in range(R):
            lanes=[fn(t,r) if t<nthreads else None for t in range(w0,w0+64)]
            if all(a is None for a in lanes): continue
            tot+=conflicts(lanes,write); ideal+= ((2 if write else 1) if F32 else (4 if write else 2))
    return tot,ideal
G1,G2,G3=N//R1,N//R2,N//R3
PMAX=65 if F32 else 17
best={}
for p in range(0,PMAX):
    pitch1=G1+p
    w=total(lambda j,r: r*pitch1+j, G1, R1, True)
    rd=total(lambda j,r: (j%R1)*pitch1+(j//R1)+R3*r, G2, R2, False)
    print('ex1 p',p,'write',w,'read',rd)
for p in range(0,PMAX):
    pitch2=G3+p
    w=total(lambda j,r: (j//R1)*pitch2+(j%R1)+R1*r, G2, R2, True)
    rd=total(lambda j,r: j+r*pitch2, G3, R3, False)
    print('ex2 p',p,'write',w,'read',rd)
for p in range(0,PMAX):
    pitch3=G3+p
    w=total(lambda j,r: r*pitch3+j, G3, R3, True)
    rd=total(lambda j,r: (j%R3)*pitch3+(j//R3)+R1*r, G2, R2, False)
    print('ex3 p',p,'write',w,'read',rd)
for p in range(0,PMAX):
    pitch4=G1+p
    w=total(lambda j,r: (j//R3)*pitch4+(j%R3)+R3*r, G2, R2, True)
    rd=total(lambda j,r: j+r*pitch4, G1, R1, False)
    print('ex4 p',p,'write',w,'read',rd)
