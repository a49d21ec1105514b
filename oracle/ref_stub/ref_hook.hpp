// ORACLE/_ref - TEST INFRASTRUCTURE ONLY. The strategy sources draw their random numbers with torch::multinomial / torch::randn_like / torch::randn; the Makefile
// pipes them through sed so that those three calls go through here: the real libtorch function runs, and its result is appended to a log that the shim hands to
// the test, which replays the same draws into the product's strategies (the GPU's generator produces different streams, so parity is "same draws -> same state").
#pragma once
#include <torch/torch.h>
#include <string>
#include <utility>
#include <vector>
namespace ref_hook {
    inline std::vector<std::pair<std::string, torch::Tensor>>& log() {
        static std::vector<std::pair<std::string, torch::Tensor>> l;
        return l;
    }
    inline torch::Tensor multinomial(const torch::Tensor& w, int64_t n, bool replacement) {
        auto r = torch::multinomial(w, n, replacement);
        log().emplace_back("multinomial", r.clone());
        return r;
    }
    inline torch::Tensor randn_like(const torch::Tensor& t) {
        auto r = torch::randn_like(t);
        log().emplace_back("randn_like", r.clone());
        return r;
    }
    inline torch::Tensor randn(torch::IntArrayRef size, const torch::TensorOptions& o) {
        auto r = torch::randn(size, o);
        log().emplace_back("randn", r.clone());
        return r;
    }
} // namespace ref_hook
