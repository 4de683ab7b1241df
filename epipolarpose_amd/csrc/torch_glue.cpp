// Host-side glue between torch's C++ autograd engine and the C ABI of libepipolar_hip (include/epipolar_hip.h).
//
// Not a kernel and not a second implementation: every function here only allocates outputs, forwards raw device
// pointers to an `epi_*` entry point on the current HIP stream, and tells autograd what to save.  It exists because the
// network calls the fused BatchNorm 53 times per step in each direction, and a Python autograd.Function + ctypes call costs
// ~35 us of host time per call -- at batch 32 the step had become host-bound (tools/host_profile.py).  The Python classes
// in models/fused.py keep the module interface (parameters, buffers, state_dict) and call into this extension.
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>

#include "../../include/epipolar_hip.h"

namespace {

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

inline epi_stream_t current_stream(const Tensor& t) {
    return reinterpret_cast<epi_stream_t>(c10::hip::getCurrentHIPStream(t.device().index()).stream());
}

inline void check(int status, const char* what) {
    TORCH_CHECK(status == EPI_OK, what, " failed: ", epi_status_string(status));
}

inline bool nhwc_bf16(const Tensor& t) {
    return t.scalar_type() == at::kBFloat16 && t.dim() == 4 && t.is_contiguous(at::MemoryFormat::ChannelsLast);
}

// flags (CPU int32[2], owned by the module): [0] sums_ws still holds a forward's sums, [1] bwd_sums still holds a
// backward's sums -- the accumulator hand-over protocol of epi_bn_act_fwd / epi_bn_act_bwd (see include/epipolar_hip.h)
struct BnAct : public torch::autograd::Function<BnAct> {
    static Tensor forward(AutogradContext* ctx, Tensor x, Tensor weight, Tensor bias, c10::optional<Tensor> residual_opt, Tensor running_mean,
                          Tensor running_var, Tensor num_batches, Tensor sums_ws, Tensor bwd_sums, Tensor flags, bool training,
                          double momentum, double eps, bool relu) {
        TORCH_CHECK(x.is_cuda(), "FusedBatchNormAct: input must live on the GPU (no CPU fallback in epipolarpose_amd)");
        TORCH_CHECK(nhwc_bf16(x), "FusedBatchNormAct: x must be channels_last bf16");
        const bool has_res = residual_opt.has_value() && residual_opt->defined();
        const Tensor residual = has_res ? *residual_opt : Tensor();
        if (has_res) TORCH_CHECK(nhwc_bf16(residual) && residual.sizes() == x.sizes(), "FusedBatchNormAct: residual layout");
        const int64_t B = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3);
        int* fl = flags.data_ptr<int>();
        if (training) {
            if (fl[0]) sums_ws.zero_();
            fl[0] = 1;
            fl[1] = 0;
        }
        Tensor y = at::empty_like(x);
        Tensor stats = at::empty({4 * C}, weight.options().dtype(at::kFloat));
        float* sp = stats.data_ptr<float>();
        check(epi_bn_act_fwd(x.data_ptr(), has_res ? residual.data_ptr() : nullptr, B * H * W, (int)C, weight.data_ptr<float>(),
                             bias.data_ptr<float>(), (float)eps, (float)momentum, training ? 1 : 0, relu ? 1 : 0,
                             running_mean.data_ptr<float>(), running_var.data_ptr<float>(), reinterpret_cast<long long*>(num_batches.data_ptr<int64_t>()), sp, sp + C,
                             sp + 2 * C, sums_ws.data_ptr<float>(), training ? bwd_sums.data_ptr<float>() : nullptr, y.data_ptr(),
                             current_stream(x)),
              "epi_bn_act_fwd");
        ctx->saved_data["training"] = training;
        if (training) {
            ctx->saved_data["relu"] = relu;
            ctx->saved_data["has_res"] = has_res;
            ctx->saved_data["sums_ws"] = sums_ws;
            ctx->saved_data["bwd_sums"] = bwd_sums;
            ctx->saved_data["flags"] = flags;
            ctx->save_for_backward({x, (relu && has_res) ? y : Tensor(), stats, weight});
        }
        return y;
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        TORCH_CHECK(ctx->saved_data["training"].toBool(),
                    "FusedBatchNormAct: backward through inference-mode statistics is not supported");
        const auto saved = ctx->get_saved_variables();
        const Tensor &x = saved[0], &y = saved[1], &stats = saved[2], &weight = saved[3];
        const bool relu = ctx->saved_data["relu"].toBool(), has_res = ctx->saved_data["has_res"].toBool();
        Tensor sums_ws = ctx->saved_data["sums_ws"].toTensor(), bwd_sums = ctx->saved_data["bwd_sums"].toTensor();
        Tensor flags = ctx->saved_data["flags"].toTensor();
        Tensor dy = grads[0];
        if (!nhwc_bf16(dy)) dy = dy.to(at::kBFloat16).contiguous(at::MemoryFormat::ChannelsLast);
        const int64_t B = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3);
        int* fl = flags.data_ptr<int>();
        // bwd_sums was cleared by this layer's forward pass; a second backward without a forward in between must not
        // touch it again -- the first one's parameter gradients may alias it
        Tensor sums = fl[1] ? at::zeros({2 * C}, stats.options()) : bwd_sums;
        Tensor dx = at::empty_like(x);
        Tensor dres = has_res ? at::empty_like(x) : Tensor();
        const float* sp = stats.data_ptr<float>();
        check(epi_bn_act_bwd(dy.data_ptr(), x.data_ptr(), y.defined() ? y.data_ptr() : nullptr, B * H * W, (int)C, weight.data_ptr<float>(),
                             sp, sp + C, sp + 2 * C, relu ? 1 : 0, sums.data_ptr<float>(), dx.data_ptr(),
                             has_res ? dres.data_ptr() : nullptr, sums_ws.data_ptr<float>(), current_stream(x)),
              "epi_bn_act_bwd");
        fl[0] = 0;
        fl[1] = 1;
        return {dx, sums.slice(0, C, 2 * C), sums.slice(0, 0, C), dres, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(),
                Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

Tensor bn_act(Tensor x, Tensor weight, Tensor bias, c10::optional<Tensor> residual, Tensor running_mean, Tensor running_var,
              Tensor num_batches, Tensor sums_ws, Tensor bwd_sums, Tensor flags, bool training, double momentum, double eps, bool relu) {
    return BnAct::apply(x, weight, bias, residual, running_mean, running_var, num_batches, sums_ws,
                        bwd_sums, flags, training, momentum, eps, relu);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "torch-autograd glue over the libepipolar_hip C ABI (no compute of its own)";
    m.def("bn_act", &bn_act, "fused BatchNorm (+residual) (+ReLU), NHWC bf16, autograd-aware");
    m.def("abi_version", []() { return epi_version(); });
}
