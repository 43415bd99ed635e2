// A C++ caller of the multi-device group API (include/dinov2_hip.h): two ranks on device 0, batch 5 split 3 + 2, logits of
// the group equal the logits of one session.  Used by tests/test_gpu_group.py::test_group_from_cpp.
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
    const bool same = std::memcmp(lg.data(), ls.data(), lg.size() * sizeof(float)) == 0;
    printf("group of %d, broadcast %.2f ms, logits %s\n", dinov2_hip_group_size(grp), dinov2_hip_group_broadcast_ms(grp), same ? "equal" : "DIFFER");
    dinov2_hip_group_free(grp);
    if (same) printf("GROUP_OK\n");
    return same ? 0 : 1;
}
