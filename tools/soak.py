import sys; sys.path.insert(0,'/root/repo')
import numpy as np, fluidx12_amd as fx
for storage in ("fp32","fp16"):
    f=fx.Fluid(); assert f.Init(640,480,(128,128,128),storage=storage)
    view,proj,eye=fx.default_camera(640,480)
    for k in range(3000):
        f.UpdateFrame(np.float32(2.0/128),k%3,view,proj,eye); f.Simulate(k%3)
        if k%500==499:
            f.ClearRenderTarget(); f.Render(k%3, fx.Fluid.OPTIMIZED, to_target=True); f.Synchronize()
            v=f.download(fx.FIELD_VELOCITY); c=f.download(fx.FIELD_COLOR); p=f.download(fx.FIELD_PRESSURE)
            print(storage,k+1,'finite',np.isfinite(v).all() and np.isfinite(c).all() and np.isfinite(p).all(),'|v|max %.3f'%np.abs(v).max(),'alpha mean %.4f'%c[...,3].mean(),'p range %.3f %.3f'%(p.min(),p.max()), 'img alpha max', int(f.download(fx.FIELD_TARGET)[...,3].max()))
