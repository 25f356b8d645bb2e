"""Where the raw-log pipeline waits (developer probe, not a bench): per blob the time the host spends blocked in
kta_kafka_blob_acquire (the ring is full: the GPU side is the bound) and in kta_kafka_blob_submit (header index + launches), per codec."""
import ctypes as C, numpy as np, time, sys
sys.path.insert(0,'/root/repo')
import bench, kafka_topic_analyzer_amd as kta
from kafka_topic_analyzer_amd import _native as N
lib=N.load(); spec,_=kta.synth_preset("c4")
nc=1_000_000; rpb=60
def make(codec):
    enc = codec if codec in (0,2,3) else 0x100
    cl=C.c_uint64()
    lib.kta_kafka_encode_synth_host_ex(C.byref(spec),0,nc,rpb,enc,None,0,C.byref(cl))
    cbuf=np.zeros(cl.value+128,np.uint8)
    lib.kta_kafka_encode_synth_host_ex(C.byref(spec),0,nc,rpb,enc,cbuf.ctypes.data,cl.value,C.byref(cl))
    if codec in (1,4):
        z=bench._recompress_batches(lib,cbuf[:cl.value].tobytes(),codec)
        cbuf=np.zeros(len(z)+128,np.uint8); cbuf[:len(z)]=np.frombuffer(z,np.uint8); cl.value=len(z)
    return cbuf, cl.value
for codec,name in [(0,"plain-1M"),(2,"snappy"),(1,"gzip"),(4,"zstd"),(3,"lz4")]:
    log, n = make(codec)
    h=kta.HipMetricHandler(256)
    h._check(lib.kta_kafka_configure(h._ctx,0,3))
    ta=ts=0.0; tot=0; t0=None
    for k in range(3+24):
        if k==3: h.sync(); t0=time.perf_counter(); ta=ts=0.0
        ptr,cap=C.c_void_p(),C.c_uint64()
        a=time.perf_counter()
        h._check(lib.kta_kafka_blob_acquire(h._ctx,C.byref(ptr),C.byref(cap)))
        b=time.perf_counter()
        use=min(n, cap.value)
        if k<3: C.memmove(ptr, log.ctypes.data, use)
        st=N.KtaKafkaIndexStats()
        h._check(lib.kta_kafka_blob_submit(h._ctx,use,k%256,C.byref(st)))
        c=time.perf_counter()
        ta+=b-a; ts+=c-b
        if k>=3: tot+=st.bytes_consumed
    h.sync(); dt=time.perf_counter()-t0
    print(name, "GB/s", round(tot/dt/1e9,1), "per blob ms", round(dt/24*1e3,3), "acquire(wait) ms", round(ta/24*1e3,3), "submit(host) ms", round(ts/24*1e3,3), "blob MB", round(n/1e6,1))
    h.close()
