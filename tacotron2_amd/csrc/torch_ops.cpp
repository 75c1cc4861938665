// TORCH_LIBRARY registration of the loop-level entry points of libtacotron2_amd.so ("What the native layer
// exports"): the same C ABI (include/tacotron2_amd.h), visible to the PyTorch dispatcher as torch.ops.tacotron2_amd.*.
//
//   encoder_lstm_fwd / encoder_lstm_bwd     the bi-LSTM of Encoder.forward over all Ti steps, both directions
//                                           (reference model.py:181-188)                 t2amd_lstm_seq_{fwd,bwd}2_f32
//   decoder_train_fwd / decoder_train_bwd   the teacher-forced decoder loop and its BPTT (model.py:405-411 around
//                                           :340-379, and its autograd backward)          t2amd_decoder_train_{fwd,bwd}_loop_f32
//   decoder_infer_steps                     a run of free-running decode steps incl. the stop test (model.py:435-449)
//                                                                                         t2amd_decoder_infer_steps_f32
//   decoder_infer_persistent                the whole single-utterance decode as one persistent launch
//                                                                                         t2amd_decoder_infer_persistent_f32
//   conv_gemm / conv_gemm16                 the dense products of the encoder / postnet convolution stacks and of the
//                                           deferred weight gradients (model.py:141-146, 174-175)   t2amd_gemm_f32 / t2amd_gemm16_tn
//   wgrad_gemm16                            1..4 K-major weight-gradient products dW = dG^T . X in one launch (desc = that many
//                                           t2amd_gemm16_desc back to back; model.py:352-371, layers.py:37-39 under autograd)
//                                                                                         t2amd_gemm16_kk_group
//
// Calling convention of every op:  op(Tensor desc, Tensor[] reads, Tensor(a!)[] writes) -> ()
//   desc    a CPU uint8 tensor holding the bytes of the entry point's descriptor struct (raw device pointers and sizes,
//           exactly what the C ABI takes; tacotron2_amd/native.py builds it with ctypes from the same header);
//   reads   every tensor the descriptor's const pointers point into;
//   writes  every tensor it writes (schema-annotated as mutated: functionalisation / torch.compile see the aliasing).
// The ops are registered for the CUDA dispatch key (= HIP on ROCm builds of torch) and enqueue on the current HIP stream
// of the tensors' device; nothing allocates; an error code of the C ABI becomes a RuntimeError carrying t2amd_last_error().
// The CPU key is registered too and always raises: there is no CPU compute path (the CPU oracle lives in oracle/, tests only).
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include "../../include/tacotron2_amd.h"

namespace {

template <typename Desc>
const Desc* as_desc(const at::Tensor& t, const char* op) {
    TORCH_CHECK(t.device().is_cpu() && t.scalar_type() == at::kByte && t.is_contiguous(),
                "tacotron2_amd::", op, ": desc must be a contiguous CPU uint8 tensor");
    TORCH_CHECK(static_cast<size_t>(t.numel()) == sizeof(Desc), "tacotron2_amd::", op, ": descriptor of ", t.numel(),
                " bytes, the library's struct has ", sizeof(Desc), " (ABI mismatch)");
    return reinterpret_cast<const Desc*>(t.data_ptr());
}

// the current HIP stream of the one device all tensor operands live on
void* launch_stream(at::TensorList reads, at::TensorList writes, const char* op) {
    int dev = -1;
    for (const auto* lst : {&reads, &writes})
        for (const at::Tensor& t : *lst) {
            if (!t.defined()) continue;
            TORCH_CHECK(t.is_cuda(), "tacotron2_amd::", op, ": operand on ", t.device(), "; the engine runs on the MI355X only");
            TORCH_CHECK(dev < 0 || dev == t.get_device(), "tacotron2_amd::", op, ": operands on different devices");
            dev = t.get_device();
        }
    TORCH_CHECK(dev >= 0, "tacotron2_amd::", op, ": no device operand");
    return c10::hip::getCurrentHIPStream(static_cast<c10::DeviceIndex>(dev)).stream();
}

void check(int rc, const char* op) {
    TORCH_CHECK(rc == T2AMD_OK, "tacotron2_amd::", op, " failed (code ", rc, "): ", t2amd_last_error());
}

void encoder_lstm_fwd(const at::Tensor& d0, const at::Tensor& d1, at::TensorList reads, at::TensorList writes) {
    check(t2amd_lstm_seq_fwd2_f32(as_desc<t2amd_lstm_seq>(d0, "encoder_lstm_fwd"), as_desc<t2amd_lstm_seq>(d1, "encoder_lstm_fwd"),
                                  launch_stream(reads, writes, "encoder_lstm_fwd")), "encoder_lstm_fwd");
}
void encoder_lstm_bwd(const at::Tensor& d0, const at::Tensor& d1, at::TensorList reads, at::TensorList writes) {
    check(t2amd_lstm_seq_bwd2_f32(as_desc<t2amd_lstm_seq>(d0, "encoder_lstm_bwd"), as_desc<t2amd_lstm_seq>(d1, "encoder_lstm_bwd"),
                                  launch_stream(reads, writes, "encoder_lstm_bwd")), "encoder_lstm_bwd");
}
void decoder_train_fwd(const at::Tensor& d, at::TensorList reads, at::TensorList writes) {
    check(t2amd_decoder_train_fwd_loop_f32(as_desc<t2amd_dec_train>(d, "decoder_train_fwd"), launch_stream(reads, writes, "decoder_train_fwd")),
          "decoder_train_fwd");
}
void decoder_train_bwd(const at::Tensor& d, at::TensorList reads, at::TensorList writes) {
    check(t2amd_decoder_train_bwd_loop_f32(as_desc<t2amd_dec_train_bwd>(d, "decoder_train_bwd"), launch_stream(reads, writes, "decoder_train_bwd")),
          "decoder_train_bwd");
}
void decoder_infer_steps(const at::Tensor& d, at::TensorList reads, at::TensorList writes) {
    check(t2amd_decoder_infer_steps_f32(as_desc<t2amd_dec_infer>(d, "decoder_infer_steps"), launch_stream(reads, writes, "decoder_infer_steps")),
          "decoder_infer_steps");
}
void decoder_infer_persistent(const at::Tensor& d, at::TensorList reads, at::TensorList writes) {
    check(t2amd_decoder_infer_persistent_f32(as_desc<t2amd_dec_persist>(d, "decoder_infer_persistent"),
                                             launch_stream(reads, writes, "decoder_infer_persistent")), "decoder_infer_persistent");
}
void conv_gemm(const at::Tensor& d, at::TensorList reads, at::TensorList writes) {
    check(t2amd_gemm_f32(as_desc<t2amd_gemm_desc>(d, "conv_gemm"), launch_stream(reads, writes, "conv_gemm")), "conv_gemm");
}
void conv_gemm16(const at::Tensor& d, at::TensorList reads, at::TensorList writes) {
    check(t2amd_gemm16_tn(as_desc<t2amd_gemm16_desc>(d, "conv_gemm16"), launch_stream(reads, writes, "conv_gemm16")), "conv_gemm16");
}
void wgrad_gemm16(const at::Tensor& d, at::TensorList reads, at::TensorList writes) {
    TORCH_CHECK(d.device().is_cpu() && d.scalar_type() == at::kByte && d.is_contiguous() && d.numel() > 0 &&
                    d.numel() % sizeof(t2amd_gemm16_desc) == 0 && d.numel() / sizeof(t2amd_gemm16_desc) <= 4,
                "tacotron2_amd::wgrad_gemm16: desc must hold 1..4 t2amd_gemm16_desc (", sizeof(t2amd_gemm16_desc), " bytes each) as CPU uint8");
    check(t2amd_gemm16_kk_group(reinterpret_cast<const t2amd_gemm16_desc*>(d.data_ptr()),
                                static_cast<int>(d.numel() / sizeof(t2amd_gemm16_desc)), launch_stream(reads, writes, "wgrad_gemm16")),
          "wgrad_gemm16");
}

[[noreturn]] void no_cpu_path() {
    TORCH_CHECK(false, "tacotron2_amd: the engine runs on the MI355X only; there is no CPU compute path "
                       "(the CPU oracle lives in oracle/ for tests)");
}
void cpu1(const at::Tensor&, at::TensorList, at::TensorList) { no_cpu_path(); }
void cpu2(const at::Tensor&, const at::Tensor&, at::TensorList, at::TensorList) { no_cpu_path(); }

}  // namespace

TORCH_LIBRARY(tacotron2_amd, m) {
    m.def("encoder_lstm_fwd(Tensor desc_fwd, Tensor desc_rev, Tensor[] reads, Tensor(a!)[] writes) -> ()");
    m.def("encoder_lstm_bwd(Tensor desc_fwd, Tensor desc_rev, Tensor[] reads, Tensor(a!)[] writes) -> ()");
    m.def("decoder_train_fwd(Tensor desc, Tensor[] reads, Tensor(a!)[] writes) -> ()");
    m.def("decoder_train_bwd(Tensor desc, Tensor[] reads, Tensor(a!)[] writes) -> ()");
    m.def("decoder_infer_steps(Tensor desc, Tensor[] reads, Tensor(a!)[] writes) -> ()");
    m.def("decoder_infer_persistent(Tensor desc, Tensor[] reads, Tensor(a!)[] writes) -> ()");
    m.def("conv_gemm(Tensor desc, Tensor[] reads, Tensor(a!)[] writes) -> ()");
    m.def("conv_gemm16(Tensor desc, Tensor[] reads, Tensor(a!)[] writes) -> ()");
    m.def("wgrad_gemm16(Tensor desc, Tensor[] reads, Tensor(a!)[] writes) -> ()");
    m.def("abi_version() -> int", []() -> int64_t { return t2amd_abi_version(); });
}

TORCH_LIBRARY_IMPL(tacotron2_amd, CUDA, m) {          // the CUDA dispatch key is HIP on ROCm builds of torch
    m.impl("encoder_lstm_fwd", encoder_lstm_fwd);
    m.impl("encoder_lstm_bwd", encoder_lstm_bwd);
    m.impl("decoder_train_fwd", decoder_train_fwd);
    m.impl("decoder_train_bwd", decoder_train_bwd);
    m.impl("decoder_infer_steps", decoder_infer_steps);
    m.impl("decoder_infer_persistent", decoder_infer_persistent);
    m.impl("conv_gemm", conv_gemm);
    m.impl("conv_gemm16", conv_gemm16);
    m.impl("wgrad_gemm16", wgrad_gemm16);
}

TORCH_LIBRARY_IMPL(tacotron2_amd, CPU, m) {
    m.impl("encoder_lstm_fwd", cpu2);
    m.impl("encoder_lstm_bwd", cpu2);
    m.impl("decoder_train_fwd", cpu1);
    m.impl("decoder_train_bwd", cpu1);
    m.impl("decoder_infer_steps", cpu1);
    m.impl("decoder_infer_persistent", cpu1);
    m.impl("conv_gemm", cpu1);
    m.impl("conv_gemm16", cpu1);
    m.impl("wgrad_gemm16", cpu1);
}
