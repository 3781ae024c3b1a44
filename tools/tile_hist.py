import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ddcmd_amd
from ddcmd_amd.martini import MartiniHIP
s = ddcmd_amd.make_water_setup(50)
m = MartiniHIP(s); m.eval_forces(); m.step(40)
# tile occupancy histogram via cell_start_o is internal; approximate from positions
d = m.download()
L = s.h[0]; n = int(np.floor(L/(0.5*(s.rmax+s.deltaR)))); c = L/n
ix = [np.clip(np.floor((d['r'][k]+L/2)/c).astype(int),0,n-1)+4 for k in range(3)]
t = (ix[2]//4)*10000 + (ix[1]//4)*100 + ix[0]//4
cnt = np.bincount(np.unique(t, return_inverse=True)[1])
print('tiles', len(cnt), 'mean', cnt.mean(), 'min', cnt.min(), 'max', cnt.max(), 'frac>256', (cnt>256).mean(), 'frac>320', (cnt>320).mean(), 'hist', np.histogram(cnt, bins=[0,64,128,192,256,320,384,512])[0])
