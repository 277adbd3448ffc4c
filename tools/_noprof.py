import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, aria_slam_amd as A
dev=torch.device("cuda",0); W,H,NF,NP=640,480,2000,2048; B=2*NP
host=torch.empty((B,H,W),dtype=torch.uint8); A.synth_sequence(1,NP,W,H,out=host.numpy()); images=host.to(dev)
stream=torch.cuda.current_stream(dev).cuda_stream
for chunk in (64,128,256,512,1024):
    ext=A.OrbHipExtractor(max_features=NF,stream=stream,max_width=W,max_height=H,max_batch=chunk); mat=A.HipMatcher(stream=stream)
    cap=ext.kp_capacity()
    kps=torch.empty((B,cap,24),dtype=torch.uint8,device=dev); desc=torch.zeros((B,cap,32),dtype=torch.uint8,device=dev); counts=torch.zeros((B,),dtype=torch.int32,device=dev)
    matches=torch.empty((B,cap,12),dtype=torch.uint8,device=dev); nm=torch.zeros((B,),dtype=torch.int32,device=dev)
    def step():
        ext.extract_batch_device(images,B,W,H,kps,desc,counts,cap)
        mat.match_batch_device(desc.data_ptr()+cap*32,counts.data_ptr()+4,desc,counts,B-1,cap*32,0.75,matches.data_ptr()+cap*12,nm.data_ptr()+4,cap)
    for prof in (0,1):
        ext.set_profiling(prof); mat.set_profiling(prof)
        step(); torch.cuda.synchronize()
        t=time.perf_counter()
        for _ in range(3): step()
        torch.cuda.synchronize(); dt=time.perf_counter()-t
        print("chunk",chunk,"prof",prof,"us/frame %.2f"%(1e6*dt/(3*B)), "f/s %.0f"%(3*B/dt))
        ext.get_profile(); mat.get_profile()
    ext.close(); mat.close()
