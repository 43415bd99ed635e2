// A C++ caller of the multi-device group API (include/dinov2_hip.h): two ranks on device 0, batch 5 split 3 + 2, logits of
// the group equal the logits of one session, blocking and with two jobs in flight (submit / wait, page-locked buffers).  Used by tests/test_gpu_group.py::test_group_from_cpp.
#include <cstdio>
#include <cstring>
#include <vector>

#include "dinov2_hip.h"

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    char err[256] = {0};
    dinov2_hip_group_opts go;
    dinov2_hip_default_group_opts(&go);
    const int32_t devs[2] = {0, 0};
    go.n_devices = 2;
    go.devices = devs;
    dinov2_hip_group* grp = nullptr;
    if (dinov2_hip_group_create(argv[1], &go, &grp, err, sizeof err) != DINOV2_HIP_OK) { fprintf(stderr, "group: %s\n", err); return 1; }
    dinov2_hip_hparams hp;
    dinov2_hip_model_hparams(dinov2_hip_group_model(grp, 0), &hp);
    const int B = 5, S = 70;
    std::vector<float> img((size_t)B * 3 * S * S);
    unsigned s = 12345;
    for (auto& v : img) { s = s * 1664525u + 1013904223u; v = (float)((s >> 8) & 0xffff) / 32768.0f - 1.0f; }
    std::vector<float> lg((size_t)B * hp.num_classes), ls((size_t)B * hp.num_classes);
    dinov2_hip_input in{img.data(), B, S, S, DINOV2_HIP_RGB_CHW, 0};
    dinov2_hip_output out{};
    out.logits = lg.data();
    if (dinov2_hip_group_predict(grp, &in, &out, DINOV2_HIP_CLASSIFY, err, sizeof err) != DINOV2_HIP_OK) { fprintf(stderr, "predict: %s\n", err); return 1; }
    dinov2_hip_session* sess = nullptr;
    if (dinov2_hip_session_create(dinov2_hip_group_model(grp, 0), nullptr, &sess, err, sizeof err) != DINOV2_HIP_OK) return 1;
    out.logits = ls.data();
    if (dinov2_hip_predict(sess, &in, &out, DINOV2_HIP_CLASSIFY, err, sizeof err) != DINOV2_HIP_OK) return 1;
    dinov2_hip_session_free(sess);
    // the same batch twice through submit / wait with both jobs in flight (one host thread), images and results in page-locked
    // buffers from the library's own allocator: the serving loop of INTEGRATION.md section 7
    bool pipelined = true;
    {
        float* pin_in = (float*)dinov2_hip_host_alloc(img.size() * sizeof(float));
        float* pin_out[2] = {(float*)dinov2_hip_host_alloc(lg.size() * sizeof(float)), (float*)dinov2_hip_host_alloc(lg.size() * sizeof(float))};
        if (!pin_in || !pin_out[0] || !pin_out[1]) { fprintf(stderr, "host_alloc failed\n"); return 1; }
        std::memcpy(pin_in, img.data(), img.size() * sizeof(float));
        dinov2_hip_input pin{pin_in, B, S, S, DINOV2_HIP_RGB_CHW, 0};
        int64_t t[2];
        for (int k = 0; k < 2; ++k) {
            dinov2_hip_output po{};
            po.logits = pin_out[k];
            if (dinov2_hip_group_submit(grp, &pin, &po, DINOV2_HIP_CLASSIFY, &t[k], err, sizeof err) != DINOV2_HIP_OK) { fprintf(stderr, "submit: %s\n", err); return 1; }
        }
        int64_t extra;
        dinov2_hip_output po{};
        po.logits = pin_out[0];
        pipelined = dinov2_hip_group_submit(grp, &pin, &po, DINOV2_HIP_CLASSIFY, &extra, err, sizeof err) != DINOV2_HIP_OK;  // a third one is refused
        for (int k = 0; k < 2; ++k) {
            if (dinov2_hip_group_wait(grp, t[k], err, sizeof err) != DINOV2_HIP_OK) { fprintf(stderr, "wait: %s\n", err); return 1; }
            pipelined = pipelined && std::memcmp(pin_out[k], ls.data(), ls.size() * sizeof(float)) == 0;
        }
        dinov2_hip_host_free(pin_in);
        dinov2_hip_host_free(pin_out[0]);
        dinov2_hip_host_free(pin_out[1]);
    }
    const bool same = pipelined && std::memcmp(lg.data(), ls.data(), lg.size() * sizeof(float)) == 0;
    printf("group of %d, broadcast %.2f ms, logits %s\n", dinov2_hip_group_size(grp), dinov2_hip_group_broadcast_ms(grp), same ? "equal" : "DIFFER");
    dinov2_hip_group_free(grp);
    if (same) printf("GROUP_OK\n");
    return same ? 0 : 1;
}
