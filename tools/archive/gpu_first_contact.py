# first GPU contact: one KITTI frame through the HIP path vs the fxp oracle
import sys, os, time
sys.path.insert(0,'tests'); sys.path.insert(0,'patchwork-plusplus_amd/python')
import numpy as np, conftest, oracle_lib as ol, pwpp_hip
O=ol.restatement()
h=pwpp_hip.Handle()
print("fxp shift", h.fxp_shift())
for k in range(6):
    a=conftest.load_kitti(k)
    ref=ol.Estimator(O,arith=ol.ARITH_FXP).run(a)
    h.estimate_ground_batch([a], mode=pwpp_hip.MODE_FRESH)
    ng,nn,npch=h.counts(0)
    g=np.sort(h.ground_indices(0)); n=np.sort(h.nonground_indices(0))
    rec=h.patch_records(0)
    ok_sets=np.array_equal(g,np.sort(ref.ground_idx)) and np.array_equal(n,np.sort(ref.nonground_idx))
    print(k,"counts",(ng,nn,npch),"oracle",(len(ref.ground_idx),len(ref.nonground_idx),len(ref.centers)),"sets",ok_sets,"time_us",h.time_us())
    orr=ref.records
    if len(rec)==len(orr):
        for fld in ("bin","n_points","n_ground","n_nonground","decision","mean","normal","sv","d"):
            eq=np.array_equal(rec[fld],orr[fld],equal_nan=True)
            if not eq:
                bad=np.nonzero(~np.all(np.atleast_2d((rec[fld]==orr[fld]).T).T.reshape(len(rec),-1),axis=1))[0]
                print("   field",fld,"differs in",len(bad),"patches; first",bad[:5], rec[fld][bad[:2]], orr[fld][bad[:2]])
    else: print("   record count differs",len(rec),len(orr))
    st=h.state(0)
    print("   state",st.sensor_height, ref.sensor_height, list(st.elevation_thr)==list(ref.elevation_thr), list(st.flatness_thr)==list(ref.flatness_thr))
    c=h.centers(0); nr=h.normals(0)
    print("   centers eq",np.array_equal(c,ref.centers,equal_nan=True),"normals eq",np.array_equal(nr,ref.normals,equal_nan=True))
