import sys, numpy as np
sys.path.insert(0,'/root/repo')
import aria_slam_amd as A
from oracle import oracle_py as O
for (w,h,nf,seed) in [(1408,1408,4000,11),(1173,1173,2000,5),(978,978,2000,5),(1000,300,1000,2),(1408,200,1000,2)]:
    a,_=A.synth_frame_pair(seed,w,h)
    e=A.OrbHipExtractor(max_features=nf,max_width=w,max_height=h)
    try:
        f=e.extract(a)
    except Exception as ex:
        print(w,h,"ERR",ex); e.close(); continue
    k,d=O.orb_extract(a,O.default_params(nf))
    gk=f["keypoints"]
    print(w,h,"n",len(gk),len(k),"equal",gk.tobytes()==k.tobytes() and np.array_equal(f["descriptors"],d))
    if gk.tobytes()!=k.tobytes():
        for l in range(8):
            s1=set(map(tuple,np.stack([gk['x'][gk['octave']==l],gk['y'][gk['octave']==l]],1).tolist()))
            s2=set(map(tuple,np.stack([k['x'][k['octave']==l],k['y'][k['octave']==l]],1).tolist()))
            print("  level",l,"gpu",len(s1),"oracle",len(s2),"common",len(s1&s2), "only_oracle sample",sorted(s2-s1)[:3])
        p=O.default_params(nf); raw=O.build_pyramid(a,p); bl=O.blur_pyramid(raw,p,w,h)
        info=e.level_info(w,h)
        for l in range(8):
            g=e.debug_read_level(l,True,info[l][0],info[l][1])
            nd=np.argwhere(g!=bl[l])
            print("  blur level",l,"diff px",len(nd), nd[:3].tolist())
    e.close()
