import numpy as np, sys
raw=np.fromfile(sys.argv[1],dtype=np.uint64)
a=raw[:-16].reshape(-1,16,5).astype(np.int64)[:512]
used=a[:,:,0]>0
t0=a[used].min()
ends=np.array([ (a[w][used[w]][:,4].max()-t0)/100 for w in range(512) if used[w].any()])
cnt=used.sum(axis=1)
print("walker end times: min %.1f median %.1f p90 %.1f max %.1f us; patches per walker min %d max %d"%(ends.min(),np.median(ends),np.percentile(ends,90),ends.max(),cnt.min(),cnt.max()))
sw=(a[:,:,3]-a[:,:,2])[used]/100
print("sweeps per patch: median %.2f p90 %.2f max %.2f"%(np.median(sw),np.percentile(sw,90),sw.max()))
ld=(a[:,:,1]-a[:,:,0])[used]/100
print("load per patch: median %.2f p90 %.2f max %.2f"%(np.median(ld),np.percentile(ld,90),ld.max()))
