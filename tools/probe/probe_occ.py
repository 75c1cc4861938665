import sys; sys.path.insert(0, '.')
import torch
from tacotron2_amd import native as nv
nv.load()
d = nv.LstmSeq(); d.B, d.T, d.H = 64, 50, 256
cus = torch.cuda.get_device_properties(0).multi_processor_count
print("cus", cus, "bwd:", nv.lstm_seq_bwd2_batch_persistent_supported(d, 2, cus), "| fwd:", nv.lstm_seq_batch_persistent_supported(d, 2, cus))
d.B = 32
print("B=32 bwd:", nv.lstm_seq_bwd2_batch_persistent_supported(d, 2, cus))
