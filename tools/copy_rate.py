import sys; sys.path.insert(0,'/root/repo')
import torch
from photon_amd.library import PhotonLibrary
lib=PhotonLibrary(); 
for nb in (1<<28, 1<<30, 1<<31):
    print(nb>>20, "MiB", round(lib.measure_copy_gbs(nb, 5),1), "GB/s")
