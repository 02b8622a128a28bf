"""The tie path of approximatenn_amd/csrc/ann_tie.h in plain Python, beside a literal model of the reference's
sort_and_uniq (/root/reference/alg.c:224-230: do_sort, rdups, do_sort; network of compute.cl:188-203, strict `>`).

literal(L, length, key, ids, k)        the network on (key, id) pairs, kill the first of adjacent equal ids, network; first k
tie_path(L, length, key, ids, k, n)    the same answer WITHOUT sorting, for rows whose k+1 smallest distinct keys hold one run
                                       of equal distances: class bits (below / tied / above the tied distance) pushed through
                                       the comparator sequence, the tied entries following their partners' classes.  None when
                                       the row does not qualify (the kernel then runs the network).
`python tools/tie_model.py` checks tie_path against literal on 20 000 random rows; tests/test_tie_model.py runs a shorter
version of that in the CPU suite, tests/test_gpu_tie.py compares the kernel with both."""
import random
INF=float('inf')
def lg(x):
    r=0
    while x>1: x>>=1; r+=1
    return r
def pairs(L):
    lk=lg(L); npairs=8<<(lk-4 if lk>4 else 0)
    for s in range(lk):
        for ss in range(s,-1,-1):
            out=[]
            for pr in range(npairs):
                hi=(pr>>ss)<<ss; lo=pr^hi; ia=hi<<1|lo
                if ss==s: lo=(1<<ss)-lo-1
                ib=hi<<1|(1<<ss)|lo
                if ib<L: out.append((ia,ib))
            yield s,ss,out
def net(L,key,ids):
    for s,ss,ps in pairs(L):
        for ia,ib in ps:
            if key[ia]>key[ib]:
                key[ia],key[ib]=key[ib],key[ia]; ids[ia],ids[ib]=ids[ib],ids[ia]
def literal(L,length,key,ids,k):
    key=list(key); ids=list(ids)
    net(L,key,ids)
    kill=[y for y in range(length-1) if ids[y]==ids[y+1]]
    for y in kill: key[y]=INF
    net(L,key,ids)
    return ids[:k],key[:k]
def sim(L,A,B,pos):
    # A,B lists of bits len P; pos list of tracked positions
    P=1<<lg(L)
    for s,ss,ps in pairs(L):
        mask=((2<<s)-1) if ss==s else (1<<ss)
        npos=[]
        for p in pos:
            q=p^mask
            isa=((p>>ss)&1)==0
            sw=(B[q]==0) if isa else (A[q]==1)
            npos.append(q if sw else p)
        pos=npos
        A2=A[:];B2=B[:]
        for ia,ib in ps:
            assert ib==ia^mask
            A2[ia]=A[ia]&A[ib];A2[ib]=A[ia]|A[ib]
            B2[ia]=B[ia]&B[ib];B2[ib]=B[ia]|B[ib]
        A,B=A2,B2
    return pos
def tie_path(L,length,key,ids,k,n):
    P=1<<lg(L)
    if L<16 or k>P: return None
    # top k+1 distinct keys over first P entries ... (as stage 1: slots below P)
    ks=sorted(set((key[j],ids[j]) for j in range(P) if key[j]<INF))[:k+1]
    m=len(ks)
    if m<k: return None
    nv=sum(1 for j in range(P) if key[j]<INF)
    if L>P and nv>=P: return None
    eq=[t for t in range(m-1) if ks[t][0]==ks[t+1][0]]
    if not eq: return None
    t0=eq[0]; t1=eq[-1]+1
    if eq!=list(range(t0,t1)): return None
    v=ks[t0][0]
    A=[1 if key[j]>v else 0 for j in range(P)]
    B=[0 if key[j]<v else 1 for j in range(P)]
    tied=[j for j in range(P) if key[j]==v]
    nt=len(tied)
    if nt>64: return None
    c=[sum(1 for j in range(P) if ids[j]==ks[t][1]) for t in range(t0)]
    nb=sum(c)
    assert nb==sum(1 for b in B if b==0)
    if nb+nt==P and L>P: return None
    fin=sim(L,A,B,tied)
    assert sorted(fin)==list(range(nb,nb+nt)),(fin,nb,nt)
    ordr=[None]*nt
    for j,f in zip(tied,fin): ordr[f-nb]=ids[j]
    surv=[i for i in range(nt) if i==nt-1 or ordr[i]!=ordr[i+1]]
    A2=[1]*P;B2=[1]*P
    cum=0
    for t in range(t0):
        cum+=c[t]; A2[cum-1]=0;B2[cum-1]=0
    for i in surv: A2[nb+i]=0
    fin2=sim(L,A2,B2,[nb+i for i in surv])
    st=len(surv)
    assert sorted(fin2)==list(range(t0,t0+st))
    o2=[None]*st
    for i,f in zip(surv,fin2): o2[f-t0]=ordr[i]
    oi=[];od=[]
    for t in range(k):
        if t<t0: oi.append(ks[t][1]);od.append(ks[t][0])
        elif t<t0+st: oi.append(o2[t-t0]);od.append(v)
        else:
            u=t1+1+(t-t0-st); oi.append(ks[u][1]);od.append(ks[u][0])
    return oi,od


def self_check(rows=20000, seed=1):
    """tie_path == literal on random rows with forced ties; returns (rows tried, rows the tie path answered)."""
    random.seed(seed)
    tot = hit = 0
    for it in range(rows):
        L = random.choice([16, 17, 24, 32, 40, 64, 100, 128, 200, 256])
        P = 1 << lg(L)
        k = random.choice([1, 2, 3, 5, 10])
        length = min(L, max(P, k) + 1)
        n = 1000
        npts = random.choice([5, 10, 30, 100])
        pts = random.sample(range(n), npts)
        dist = {p: float(random.randint(1, 1000)) for p in pts}
        g = random.sample(pts, random.choice([2, 2, 2, 3, 4]))
        for p in g:
            dist[p] = dist[g[0]]
        ids = [n if random.random() < 0.4 else random.choice(pts) for _ in range(length)]
        key = [dist[i] if i < n else INF for i in ids]
        r = tie_path(L, length, key, ids, k, n)
        tot += 1
        if r is None:
            continue
        hit += 1
        li, ld = literal(L, length, key, ids, k)
        assert (li, ld) == (r[0], r[1]), (L, k, li, ld, r)
    return tot, hit


if __name__ == "__main__":
    print("%d rows, %d answered by the tie path, all equal to the literal network" % self_check())
