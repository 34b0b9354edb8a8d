"""Float64 numpy model of the simulation step, for STABILITY questions only (not an oracle, not used by any test).

Why it exists: tools/soak.py showed the HIP path AND the C oracle (both pinned per step to the shipped shaders) growing
without bound after ~17-19 s of simulated time (step ~450 at 32^3, ~600 at 48^3, ~1300 at 128^3 with dt = 2/Y).  This
model reproduces that in float64 and isolates the cause:

  * `run(wall=0)`            -> stable for thousands of steps (open box: flow leaves through the clamped border);
  * `run(wall=1)`            -> blows up, also with rho = 1, with a cold-started solve, and with the force switched off
                                after 200 steps (so it is not the force, the 1/0.48 gradient scale or the warm start);
  * `power_iteration()`      -> the map "project, then CSProject3D.hlsl:108-112 boundary process" applied to random
                                velocity with NO advection still grows the rms (~0.4 %/step), with 40 or 200 sweeps,
                                warm or cold: the one-sided border divergence/gradient plus the outward-only damping is
                                not a contraction, independent of how well the Poisson solve converges.

So the growth belongs to the reference's algorithm as written (CSProject3D.hlsl:44-52, 108-112), not to the sweep
schedule or to fp32; DESIGN.md section 3 records it.  Usage:

  python tools/stability_model.py run N=32 steps=800 [rho=1.0] [wall=0] [warm=0] [fsteps=200]
  python tools/stability_model.py lin 24 40 0.48 1
"""
import sys

import numpy as np
from scipy.ndimage import map_coordinates
def run(N=32, steps=1500, iters=40, rho=0.48, dt=None, warm=True, wall=True, force=True, log=50, early=False, fsteps=10**9, wallmode=0, gs=0):
    dt = 2.0/N if dt is None else dt
    u = np.zeros((3,N,N,N))   # comp (x,y,z), index [z,y,x]
    p = np.zeros((N,N,N))
    idx = np.arange(N)
    zz,yy,xx = np.meshgrid(idx,idx,idx,indexing='ij')
    pos = np.stack([(xx+0.5)/N,(yy+0.5)/N,(zz+0.5)/N])
    disp = pos - np.array([0.5,0.1,0.5])[:,None,None,None]
    basis = np.exp(-4*np.sum(disp*disp,0)/(1/16.)**2)
    mask = basis >= np.exp(-4.0)
    def sh(a,ax,d):
        # neighbor with clamp
        i = np.clip(idx+d,0,N-1)
        return np.take(a,i,axis=ax)
    for k in range(steps):
        back = pos - u*dt
        coords = [back[2]*N-0.5, back[1]*N-0.5, back[0]*N-0.5]
        un = np.stack([map_coordinates(u[c],coords,order=1,mode='nearest') for c in range(3)])
        if force and k<fsteps:
            f = np.zeros_like(un)
            f[1] = 48*4*basis
            f[0] += -disp[2]*200; f[2] += disp[0]*200
            un += np.where(mask, f*dt, 0)
        un *= max(1-0.2*dt,0)
        b = 0.5*((sh(un[0],2,1)-sh(un[0],2,-1))+(sh(un[1],1,1)-sh(un[1],1,-1))+(sh(un[2],0,1)-sh(un[2],0,-1)))
        if not warm: p[:] = 0
        for it in range(iters):
            p = (sh(p,2,1)+sh(p,2,-1)+sh(p,1,1)+sh(p,1,-1)+sh(p,0,1)+sh(p,0,-1)-b)/6
        g = np.stack([sh(p,2,1)-sh(p,2,-1), sh(p,1,1)-sh(p,1,-1), sh(p,0,1)-sh(p,0,-1)])
        un -= 0.5*g/rho
        if wall:
            q = pos*2-1
            fac = np.clip((0.97-np.abs(q))/0.03,-1,1)
            un = np.where(un*q>0, un*fac, un)
        u = un
        if k%log==log-1:
            b2 = 0.5*((sh(u[0],2,1)-sh(u[0],2,-1))+(sh(u[1],1,1)-sh(u[1],1,-1))+(sh(u[2],0,1)-sh(u[2],0,-1)))
            print(k+1,'|v|max %.4g'%np.abs(u).max(),'rms %.4g'%np.sqrt((u*u).mean()),'p mean %.4g rng %.4g %.4g'%(p.mean(),p.min(),p.max()),'div rms %.4g'%np.sqrt((b2*b2).mean()),flush=True)
            if not np.isfinite(u).all() or np.abs(u).max()>1e6: break


def power_iteration(N=24, iters=40, rho=0.48, warm=1):
    idx=np.arange(N)
    def sh(a,ax,d): return np.take(a,np.clip(idx+d,0,N-1),axis=ax)
    zz,yy,xx=np.meshgrid(idx,idx,idx,indexing='ij')
    pos=np.stack([(xx+0.5)/N,(yy+0.5)/N,(zz+0.5)/N]); q=pos*2-1
    fac=np.clip((0.97-np.abs(q))/0.03,-1,1)
    rng=np.random.default_rng(0)
    u=rng.standard_normal((3,N,N,N)); p=np.zeros((N,N,N))
    # power iteration on the (nonlinear-by-sign) project+wall map, advection = identity
    for k in range(400):
        b=0.5*((sh(u[0],2,1)-sh(u[0],2,-1))+(sh(u[1],1,1)-sh(u[1],1,-1))+(sh(u[2],0,1)-sh(u[2],0,-1)))
        if not warm: p[:]=0
        for it in range(iters):
            p=(sh(p,2,1)+sh(p,2,-1)+sh(p,1,1)+sh(p,1,-1)+sh(p,0,1)+sh(p,0,-1)-b)/6
        g=np.stack([sh(p,2,1)-sh(p,2,-1),sh(p,1,1)-sh(p,1,-1),sh(p,0,1)-sh(p,0,-1)])
        u=u-0.5*g/rho
        u=np.where(u*q>0,u*fac,u)
        n=np.sqrt((u*u).mean())
        if k%20==19: print(k+1,'rms %.4g'%n,'p mean %.4g'%p.mean(), 'p rng %.4g'%(p.max()-p.min()))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'lin':
        v = sys.argv[2:]
        power_iteration(*( [int(v[0]), int(v[1]), float(v[2]), int(v[3])][:len(v)] ))
    else:
        kw = dict(a.split('=') for a in sys.argv[2:])
        run(**{k: (float(v) if '.' in v else int(v)) for k, v in kw.items()})
