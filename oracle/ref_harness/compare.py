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
    obs_err = 0; rew_err = 0; done_bad = 0; info_err = 0
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
        rew_err = max(rew_err, abs(rew[0]-ref['reward'][t])/max(abs(ref['reward'][t]),1e-12))
        done_bad += int(done[0] != ref['done'][t])
        m = ~np.isnan(ref['info'][t])
        info_err = max(info_err, (np.abs(info[0][m]-ref['info'][t][m])/np.maximum(np.abs(ref['info'][t][m]),1e-12)).max())
    print('fields with rel err > 1e-9:', sum(1 for v in worst.values() if v[0]>1e-9), 'of', len(cols))
    for k,v in sorted(worst.items(), key=lambda kv:(kv[1][1], -kv[1][0]))[:top]:
        if v[0] > 1e-12: print('  %-40s err=%.3e t=%d mine=%r ref=%r' % (k, v[0], v[1], v[2], v[3]))
    print('max obs rel err', obs_err, 'reward', rew_err, 'info', info_err, 'done mismatches', done_bad, 'dones', int(ref['done'].sum()))
    return ref, o
if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv)>1 else 30
    which = sys.argv[3] if len(sys.argv)>3 else 'const'
    if which == 'const':
        sc = dict(name='t', steps=n, noise=True)
    elif which == 'reactor':
        rng = np.random.default_rng(5)
        acts = rng.choice([0,1,2,3,8,9,10,4,5], size=n); mags = rng.uniform(0,1,size=n)
        sc = dict(name='t', steps=n, heat_source='reactor', equilibrium=(100.0, 95.0), actions=lambda t:(int(acts[t]), float(mags[t])))
    elif which == 'default_reactor':
        sc = dict(name='t', steps=n, heat_source='reactor')
    elif which == 'scram':
        sc = dict(name='t', steps=n, heat_source='reactor', equilibrium=(100.0, 95.0), actions=lambda t:(3,1.0) if t>=5 else None)
    elif which == 'load':
        sc = dict(name='t', steps=n, noise=True, setpoints=lambda t: 100.0 - 30.0*min(1.0, t/60.0) if t<120 else 70.0 + 30.0*min(1.0,(t-120)/60.0),
                  cooling=lambda t: 25.0 + 5.0*np.sin(t/20.0))
    compare(sc, prefixes=sys.argv[2].split(',') if len(sys.argv)>2 and sys.argv[2] != 'all' else None)
