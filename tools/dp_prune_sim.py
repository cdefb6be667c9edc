"""Cell-exact simulation of the extension DP with the early stop and an admissible pruning of the live window (diagnostics;
the estimate behind DESIGN.md 3.5: how much of a wave several short jobs could share)."""
import numpy as np, sys
NEG=-10**6
match,mis,q1,e1,q2,e2=2,4,4,2,24,1
def gap(k):  # cost of a gap of length k>=1
    return min(q1+e1*k, q2+e2*k)
def run(qs, ts, S, prune=True):
    ql, tl = len(qs), len(ts)
    # arrays over t: H (after last diag), Hd (diag r-2 shifted), E (to be consumed by t+1), F, E2, F2
    Htop = np.array([-gap(t+1) for t in range(tl)], dtype=np.int64)   # H(t,-1)
    H = Htop.copy()          # unborn cells hold H(t,-1)
    F = Htop - q1 - e1       # F(t,0)
    F2 = Htop - q2 - e2
    E = np.full(tl, NEG, dtype=np.int64); E2 = E.copy()
    Hprev_shift = np.full(tl, NEG, dtype=np.int64)  # H of diag r-2 at t-1
    M = 0; mt = mq = -1
    maxw = 0; ndiag = 0; cells=0
    alive_prev = np.ones(tl, dtype=bool)
    need_fail = False
    T = np.arange(tl)
    for r in range(ql+tl-1):
        st0 = max(0, r-ql+1); en0 = min(r, tl-1)
        if st0 > en0: break
        hleft_r = -gap(r+1)            # H(-1, r)
        hleft_rm1 = 0 if r == 0 else -gap(r)  # H(-1, r-1)
        # shifted inputs
        e_in = np.empty(tl, dtype=np.int64); e_in[1:] = E[:-1]; e_in[0] = hleft_r - q1 - e1
        e2_in = np.empty(tl, dtype=np.int64); e2_in[1:] = E2[:-1]; e2_in[0] = hleft_r - q2 - e2
        hs = np.empty(tl, dtype=np.int64); hs[1:] = H[:-1]; hs[0] = hleft_r   # H(t-1, diag r-1) -> becomes diag pred next time... 
        hd = Hprev_shift.copy(); hd[0] = hleft_rm1
        live = (T >= st0) & (T <= en0)
        qi = r - T
        s = np.where(live, np.where(qs[np.clip(qi,0,ql-1)] == ts, match, -mis), 0)
        h0 = hd + s
        h = np.maximum.reduce([h0, e_in, F, e2_in, F2])
        h = np.where(live, np.maximum(h, NEG), H)
        # next gap states
        En = np.maximum(e_in, h - q1) - e1
        Fn = np.maximum(F, h - q1) - e1
        E2n = np.maximum(e2_in, h - q2) - e2
        F2n = np.maximum(F2, h - q2) - e2
        # pruning
        rows_left = ql-1-qi; cols_left = tl-1-T
        pot = match*np.minimum(rows_left, cols_left)
        if live.any():
            hm = h[live].max()
            if hm > M:
                # certification check: dead cells on this diagonal within live range
                dead = live & (h <= NEG//2)
                if dead.any():
                    # alive class maxima vs UB of dead
                    ub = (M - S - pot[dead]).max()
                    # class mins of alive maxima
                    cls = (T - st0) % 8
                    amin = min([h[live & ~dead & (cls==k)].max() if (live & ~dead & (cls==k)).any() else NEG for k in range(8)])
                    if not (amin > ub): need_fail = True
                M = hm; idx = np.where(live & (h == hm))[0]; mt = idx[0]; mq = r - mt
        if prune:
            deadnow = live & (h + pot <= M - S)
            h = np.where(deadnow, NEG, h); En = np.where(deadnow, NEG, En); Fn = np.where(deadnow, NEG, Fn)
            E2n = np.where(deadnow, NEG, E2n); F2n = np.where(deadnow, NEG, F2n)
        al = live & (h > NEG//2)
        Hprev_shift = hs
        H = h; 
        E = np.where(live, En, E); F = np.where(live, Fn, F); E2 = np.where(live, E2n, E2); F2 = np.where(live, F2n, F2)
        ndiag += 1
        if al.any():
            lo = np.where(al)[0].min(); w = en0 - lo + 1
            maxw = max(maxw, w); cells += al.sum()
        else:
            if r >= ql: break   # everything dead: stop
        # stop bound for unpruned: same rule
        if not prune and r >= ql:
            bnd = (h+pot)[live].max()
            # previous bound not tracked: approximate
            if bnd <= M and (hleft_r + match*ql) <= M: break
    return M, mt, mq, ndiag, maxw, cells, need_fail
rng=np.random.default_rng(1)
def junk(ql):
    return rng.integers(0,4,ql), rng.integers(0,4,min(ql+1000, 3*ql+80))
def real(ql, err=0.005):
    t = rng.integers(0,4,min(ql+1000,3*ql+80)); q=t[:ql].copy()
    m = rng.random(ql)<err; q[m]=(q[m]+1+rng.integers(0,3,m.sum()))%4
    return q,t
import collections
for S in (0,16,32):
    print("S",S)
    for kind,gen in (("junk",junk),("real",real)):
        for ql in (10,20,30,48,66,90,116,132):
            res=[]
            for it in range(30):
                q,t=gen(ql)
                a=run(q,t,S,True); b=run(q,t,S,False)
                res.append((a[:3]==b[:3], a[3], a[4], b[3], a[6], a[5], b[5]))
            ok=sum(x[0] for x in res); nd=np.mean([x[1] for x in res]); mw=np.max([x[2] for x in res]); mwm=np.mean([x[2] for x in res]); nd0=np.mean([x[3] for x in res]); nf=sum(x[4] for x in res)
            print(kind, "ql",ql,"same",ok,"/30 diags pruned %.0f unpruned %.0f  maxwin max %d mean %.1f  certfail %d cells %.0f vs %.0f"%(nd,nd0,mw,mwm,nf,np.mean([x[5] for x in res]),np.mean([x[6] for x in res])))
