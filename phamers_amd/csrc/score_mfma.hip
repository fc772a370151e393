// score_mfma.hip -- MFMA candidate search + exact re-rank (filled in after the exact path is
// validated on hardware; until then every model takes the exact float64 path).
#include "phk_common.h"
#include "score_model.h"

int phk_model_build_fast(phk_ctx *, phk_model *m, const double *, const double *, const double *,
                         const double *) {
    m->fast = false;
    return PHK_OK;
}
void phk_model_free_fast(phk_model *m) {
    if (m->d_Bf) (void)hipFree(m->d_Bf);
    if (m->d_colnorm) (void)hipFree(m->d_colnorm);
    if (m->d_mu32) (void)hipFree(m->d_mu32);
    if (m->d_mu64) (void)hipFree(m->d_mu64);
    m->d_Bf = nullptr; m->d_colnorm = nullptr; m->d_mu32 = nullptr; m->d_mu64 = nullptr;
}
int phk_score_fast(phk_ctx *, const phk_model *, const double *, const uint32_t *, uint64_t, int,
                   double *, uint32_t *) {
    phk_set_error("phk_score_fast: MFMA path not built");
    return PHK_ERR_UNSUPPORTED;
}
