// Compiled torch-dispatcher binding of the C ABI: `flute::qgemm_raw_simple[_hadamard]`.
//
// Role of flute/csrc/qgemm.cpp:86-198 (qgemm_raw_simple), :214-244 (qgemm_raw_simple_hadamard) and the
// registrations :246-260: same schemas (verbatim), implemented for dispatch key CUDA (HIP tensors use it on
// PyTorch-ROCm).  The function body is host glue only - validate, flatten input[..., K] -> [M, K], allocate
// D[M, N], device guard, current stream - and ONE call into libflute_amd.so (include/flute_amd.h); no device
// code lives here (plain C++ translation unit).  Round 1 did this glue in Python + ctypes: 18.9 us per eager
// call against a 6 us kernel.
#include <ATen/ATen.h>
#include <ATen/hip/HIPContext.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/library.h>

#include "../../include/flute_amd.h"

namespace {

int dtype_id(const at::Tensor& t) {
    if (t.scalar_type() == at::kHalf) return FLUTE_F16;
    if (t.scalar_type() == at::kBFloat16) return FLUTE_BF16;
    TORCH_CHECK_TYPE(false, "Only fp16 and bf16 supported currently");
    return -1;
}

// flute/ops.py:17-49 (the reference validates in its fake impl only and trusts raw pointers in the real one,
// qgemm.cpp:71-77; here both validate)
void validate(const at::Tensor& input, const at::Tensor& weight, const at::Tensor& scales, const at::Tensor& table,
              const at::Tensor& table2, const at::Tensor& workspace, int64_t num_bits, int64_t group_size) {
    TORCH_CHECK_VALUE(input.dim() >= 2 && weight.dim() == 2 && scales.dim() == 2 && table.dim() == 1 &&
                          table2.dim() == 3 && workspace.dim() == 1,
                      "flute::qgemm_raw_simple: wrong tensor ranks");
    const auto dt = input.scalar_type();
    TORCH_CHECK_TYPE(dt == at::kHalf || dt == at::kBFloat16, "Only fp16 and bf16 supported currently");
    TORCH_CHECK_TYPE(weight.scalar_type() == at::kShort && scales.scalar_type() == dt && table.scalar_type() == dt &&
                         table2.scalar_type() == at::kFloat && workspace.scalar_type() == at::kByte,
                     "flute::qgemm_raw_simple: wrong dtypes");
    TORCH_CHECK_VALUE(num_bits >= 1 && num_bits <= 8, "Unsupported num_bits value");
    const int64_t K = input.size(-1), N = scales.size(0), L = int64_t(1) << num_bits;
    TORCH_CHECK_VALUE(weight.size(1) == K && K == scales.size(1) * group_size &&
                          weight.size(0) == (int64_t)(num_bits * (N / 16.0)) && table.size(0) == L &&
                          table2.size(0) == L && table2.size(1) == L && table2.size(2) == 1,
                      "flute::qgemm_raw_simple: inconsistent shapes");
}

at::Tensor qgemm_impl(const at::Tensor& input, const at::Tensor& weight, const at::Tensor& scales,
                      const at::Tensor& table, const at::Tensor& table2, at::Tensor& workspace, int64_t num_bits,
                      int64_t group_size, int64_t hadamard_size, int64_t template_id, int64_t num_sms) {
    validate(input, weight, scales, table, table2, workspace, num_bits, group_size);
    TORCH_CHECK(weight.is_contiguous() && scales.is_contiguous() && table.is_contiguous() && table2.is_contiguous() &&
                    workspace.is_contiguous(),
                "flute::qgemm_raw_simple: weight/scales/tables/workspace must be contiguous");
    const auto dev = input.device();
    TORCH_CHECK(weight.device() == dev && scales.device() == dev && table.device() == dev && table2.device() == dev &&
                    workspace.device() == dev,
                "flute::qgemm_raw_simple: all tensors must be on the input's device");
    const int64_t K = input.size(-1), N = scales.size(0);
    const bool flat = input.dim() == 2 && input.is_contiguous();          // the decode-loop case: no view objects at all
    at::Tensor x2d = flat ? input : input.reshape({-1, K});
    if (!x2d.is_contiguous()) x2d = x2d.contiguous();
    const int64_t M = x2d.size(0);
    at::Tensor out = at::empty({M, N}, input.options());
    if (M > 0) {
        const int dt = dtype_id(input);
        if (hadamard_size != 0) {
            if (hadamard_size < 1 || (hadamard_size & (hadamard_size - 1)) || hadamard_size > (1 << 15))
                TORCH_CHECK(false, flute_strerror(FLUTE_ERR_HADAMARD_SIZE));
            TORCH_CHECK(K % hadamard_size == 0 || (M * K) % hadamard_size == 0, "shape is invalid for hadamard_size ",
                        hadamard_size);
        }
        // PyTorch-ROCm keeps DeviceType::CUDA for HIP devices: the "masquerading" guard / stream types
        const c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);                // qgemm.cpp:101 OptionalCUDAGuard
        at::Tensor scratch;
        void* scratch_ptr = nullptr;
        // decode-kernel launches rotate the activations while staging them; every other plan rotates into a
        // scratch tensor first (qgemm.cpp:201-244)
        if (hadamard_size > 1 && !flute_qgemm_hadamard_fused(dt, (int)num_bits, (int)group_size, (int)hadamard_size,
                                                             (int)M, (int)N, (int)K, (int)template_id, (int)num_sms,
                                                             (size_t)workspace.numel())) {
            scratch = at::empty_like(x2d);
            scratch_ptr = scratch.data_ptr();
        }
        const hipStream_t stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();   // qgemm.cpp:105
        const int rc = flute_qgemm_hadamard(dt, (int)num_bits, (int)group_size, (int)hadamard_size, (int)M, (int)N,
                                            (int)K, (int)weight.size(0), x2d.data_ptr(), weight.data_ptr(),
                                            out.data_ptr(), scales.data_ptr(), table.data_ptr(), table2.data_ptr(),
                                            scratch_ptr, workspace.data_ptr(), (size_t)workspace.numel(),
                                            (int)template_id, (int)num_sms, stream);
        TORCH_CHECK(rc == FLUTE_OK, flute_strerror(rc));      // RuntimeError with the reference's message prefixes
    }
    if (flat) return out;
    auto shape = input.sizes().vec();
    shape.back() = N;
    return out.reshape(shape);
}

at::Tensor qgemm_raw_simple(const at::Tensor& input, const at::Tensor& weight, const at::Tensor& scales,
                            const at::Tensor& table, const at::Tensor& table2, at::Tensor& workspace,
                            int64_t num_bits, int64_t group_size, int64_t template_id, int64_t num_sms) {
    return qgemm_impl(input, weight, scales, table, table2, workspace, num_bits, group_size, 0, template_id, num_sms);
}

at::Tensor qgemm_raw_simple_hadamard(const at::Tensor& input, const at::Tensor& weight, const at::Tensor& scales,
                                     const at::Tensor& table, const at::Tensor& table2, at::Tensor& workspace,
                                     int64_t num_bits, int64_t group_size, int64_t hadamard_size,
                                     int64_t template_id, int64_t num_sms) {
    return qgemm_impl(input, weight, scales, table, table2, workspace, num_bits, group_size, hadamard_size,
                      template_id, num_sms);
}

}  // namespace

// flute/csrc/qgemm.cpp:251-254, verbatim
TORCH_LIBRARY(flute, m) {
    m.def("qgemm_raw_simple(Tensor input, Tensor weight, Tensor scales, Tensor table, Tensor table2, "
          "Tensor(a!) workspace, int num_bits, int group_size, int template_id, int num_sms) -> Tensor");
    m.def("qgemm_raw_simple_hadamard(Tensor input, Tensor weight, Tensor scales, Tensor table, Tensor table2, "
          "Tensor(a!) workspace, int num_bits, int group_size, int hadamard_size, int template_id, int num_sms) "
          "-> Tensor");
}

// flute/csrc/qgemm.cpp:257-260
TORCH_LIBRARY_IMPL(flute, CUDA, m) {
    m.impl("qgemm_raw_simple", &qgemm_raw_simple);
    m.impl("qgemm_raw_simple_hadamard", &qgemm_raw_simple_hadamard);
}
