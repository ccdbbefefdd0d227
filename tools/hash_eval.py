import numpy as np, sys
rng=np.random.default_rng(1)
N=int(sys.argv[1]) if len(sys.argv)>1 else 30_000_000
bases=rng.integers(0,4,size=N+14,dtype=np.uint32)
# 15-mer values (LSB-first)
v=np.zeros(N,dtype=np.uint32)
for i in range(15):
    v|=bases[i:i+N]<<np.uint32(2*i)
M32=np.uint32(0xFFFFFFFF)
def mix_cur(h):
    h=h.copy()
    h+=h<<np.uint32(10); h^=h>>np.uint32(6); h+=h<<np.uint32(3); h^=h>>np.uint32(11); h+=h<<np.uint32(15)
    return h
def mix_a(h):   # 5 ops: add-shift, xor-shift, add-shift
    h=h.copy()
    h+=h<<np.uint32(10); h^=h>>np.uint32(6); h+=h<<np.uint32(15)
    return h
def mix_b(h):   # 5 ops
    h=h.copy()
    h+=h<<np.uint32(13); h^=h>>np.uint32(7); h+=h<<np.uint32(17)
    return h
def mix_c(h):   # 4 ops: two add-shifts + one xor-shift
    h=h.copy()
    h+=h<<np.uint32(9); h^=h>>np.uint32(13); h+=h<<np.uint32(16)
    return h
def mix_d(h):   # 6 ops
    h=h.copy()
    h+=h<<np.uint32(10); h^=h>>np.uint32(6); h+=h<<np.uint32(3); h^=h>>np.uint32(11)
    return h
def mix_e(h):   # mul24-based: (x_lo24*C) ^ ... 
    h=h.copy()
    lo=(h&np.uint32(0xFFFFFF)).astype(np.uint64)*np.uint64(0x9E3779)
    hi=(h>>np.uint32(6)).astype(np.uint64)*np.uint64(0x85EBCA)     # uses 24 bits (30-6)
    r=(lo.astype(np.uint32))^( (hi&np.uint64(0xFFFFFFFF)).astype(np.uint32)<<np.uint32(7))
    return r
def dig_cur(hm,c0):
    g=hm*np.uint32(0x9E3779B1)
    d0=((g>>np.uint32(16)).astype(np.uint64)*c0>>np.uint64(16)).astype(np.uint32)
    return d0,(g>>np.uint32(6))&np.uint32(1023),(g>>np.uint32(2))&np.uint32(15)
def dig_bits(hm,c0):
    d0=(((hm>>np.uint32(14))&np.uint32(0x3FFF)).astype(np.uint64)*c0>>np.uint64(14)).astype(np.uint32)
    return d0,(hm>>np.uint32(4))&np.uint32(1023),hm&np.uint32(15)
def dig_x(hm,c0):   # one xorshift to pull higher bits down first
    g=hm^(hm>>np.uint32(15))
    d0=(((g>>np.uint32(14))&np.uint32(0x3FFF)).astype(np.uint64)*c0>>np.uint64(14)).astype(np.uint32)
    return d0,(g>>np.uint32(4))&np.uint32(1023),g&np.uint32(15)
W=17
from numpy.lib.stride_tricks import sliding_window_view
def evaluate(name,mix,dig):
    h=mix(v)
    hm=sliding_window_view(h,W).min(axis=1)
    n=len(hm)
    brk=np.flatnonzero(hm[1:]!=hm[:-1])
    nrec=len(brk)+1
    # records = runs; kmers per final bucket
    starts=np.concatenate([[0],brk+1]); ends=np.concatenate([brk,[n-1]])
    lens=ends-starts+1
    d0,d1,d2=dig(hm[starts],np.uint64(68))
    # scale: simulate geometry with fewer buckets so that mean bucket ~2700 kmers: total buckets = n/2700
    nb=max(16,n//2700)
    # use combined digit index reduced: take (d0*1024+d1)*16+d2 and fold to nb by using top-level structure: coarse d0 (68) x d1 low bits
    b1=max(1,int(np.ceil(np.log2(max(1,nb/68/16)))))
    idx=(d0.astype(np.int64)*(1<<b1)+(d1&np.uint32((1<<b1)-1)))*16+d2
    cnt=np.bincount(idx,weights=lens,minlength=68*(1<<b1)*16)
    c0cnt=np.bincount(d0,weights=lens,minlength=68)
    print(f"{name:10s} kmers/rec {n/nrec:.3f} maxrun {lens.max()}  buckets {len(cnt)} mean {cnt.mean():.0f} cv {cnt.std()/cnt.mean():.3f} (poisson-ish expect {np.sqrt(9.0/cnt.mean()*1.3):.3f}) max/mean {cnt.max()/cnt.mean():.2f} frac>4096 {np.mean(cnt>4096):.4f} | coarse max/mean {c0cnt.max()/c0cnt.mean():.3f} min/mean {c0cnt.min()/c0cnt.mean():.3f}")
evaluate("cur",mix_cur,dig_cur)
evaluate("cur+bits",mix_cur,dig_bits)
evaluate("cur+x",mix_cur,dig_x)
for nm,m in (("a",mix_a),("b",mix_b),("c",mix_c),("d",mix_d),("e",mix_e)):
    evaluate(nm+"+cur",m,dig_cur)
    evaluate(nm+"+bits",m,dig_bits)
    evaluate(nm+"+x",m,dig_x)
print("---- mul24 digits")
def dig_m24(hm,c0):
    g=((hm&np.uint32(0xFFFFFF)).astype(np.uint64)*np.uint64(0x9E3779)&np.uint64(0xFFFFFFFF)).astype(np.uint32)
    d0=((g>>np.uint32(16)).astype(np.uint64)*c0>>np.uint64(16)).astype(np.uint32)
    return d0,(g>>np.uint32(6))&np.uint32(1023),(g>>np.uint32(2))&np.uint32(15)
def mix_f(h):   # 3 ops: add-shift, xor-shift
    h=h.copy()
    h+=h<<np.uint32(11); h^=h>>np.uint32(7)
    return h
def mix_g(h):   # 4 ops
    h=h.copy()
    h^=h>>np.uint32(7); h+=h<<np.uint32(11); h+=h<<np.uint32(17)
    return h
for nm,m in (("cur",mix_cur),("a",mix_a),("c",mix_c),("f",mix_f),("g",mix_g)):
    evaluate(nm+"+m24",m,dig_m24)
print("---- no adjacent shift-adds")
def mix_h(h):
    h=h.copy(); h+=h<<np.uint32(11); h^=h>>np.uint32(7); h+=h<<np.uint32(17); return h
def mix_i(h):
    h=h.copy(); h+=h<<np.uint32(13); h^=h>>np.uint32(9); h+=h<<np.uint32(15); return h
def mix_j(h):
    h=h.copy(); h+=h<<np.uint32(10); h^=h>>np.uint32(6); h+=h<<np.uint32(16); return h
for nm,m in (("h",mix_h),("i",mix_i),("j",mix_j)):
    evaluate(nm+"+m24",m,dig_m24)
