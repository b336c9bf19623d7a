import numpy as np, sys
from nuclear_sim_amd.schema import SCHEMA
from oracle.ref_harness import trace
from oracle import npo
def compare(sc, prefixes=None, P=None, top=25, seed_from_ref=False):
    cols = SCHEMA.columns()
    ref, sim = trace.run_reference(sc, cols)
    P = P or npo.Params()
    P.heat_source = 1 if sc.get('heat_source')=='reactor' else 0
    P.hs_noise_enabled = 1 if sc.get('noise') else 0
    P.dt = sc.get('dt', 1.0)
    o = npo.OraclePlants(1, P)
    f0, i0 = o.state()
    # compare initial state
    bad0 = []
    for (kind, slot, label, path), v in zip(cols, ref['state'][0]):
        if np.isnan(v): continue
        mine = f0[slot] if kind=='f64' else i0[slot]
        if mine != v: bad0.append((label, mine, v))
    print('init mismatches:', len(bad0)); 
    for b in bad0[:40]: print('   ', b)
    if sc.get('equilibrium') is not None or seed_from_ref:
        for (kind, slot, label, path), v in zip(cols, ref['state'][0]):
            if np.isnan(v): continue
            if kind=='f64': f0[slot]=v
            else: i0[slot]=int(v)
        o.set_state(f0,i0)
    worst={}
    obs_err = 0
    for t in range(sc['steps']):
        obs, rew, done, flags, info = o.step(action=ref['action'][t], magnitude=ref['magnitude'][t], setpoint=ref['setpoint'][t], noise_z=ref['noise_z'][t], cw_temp=ref['cooling'][t])
        f,i = o.state()
        for (kind, slot, label, path), v in zip(cols, ref['state'][t+1]):
            if np.isnan(v): continue
            if prefixes and not any(label.startswith(p) for p in prefixes): continue
            mine = f[slot] if kind=='f64' else i[slot]
            err = abs(mine-v)/max(abs(v),1e-12)
            if err > worst.get(label,(0,))[0]: worst[label]=(err,t,mine,v)
        e = np.abs(obs[0]-ref['obs'][t])/np.maximum(np.abs(ref['obs'][t]),1e-12)
        obs_err = max(obs_err, e.max())
    print('fields with rel err > 1e-9:', sum(1 for v in worst.values() if v[0]>1e-9), 'of', len(cols))
    for k,v in sorted(worst.items(), key=lambda kv:(kv[1][1], -kv[1][0]))[:top]:
        if v[0] > 1e-12: print('  %-40s err=%.3e t=%d mine=%r ref=%r' % (k, v[0], v[1], v[2], v[3]))
    print('max obs rel err', obs_err)
    return ref, o
if __name__ == '__main__':
    sc = dict(name='t', steps=int(sys.argv[1]) if len(sys.argv)>1 else 30, noise=True)
    compare(sc, prefixes=sys.argv[2].split(',') if len(sys.argv)>2 else None)
