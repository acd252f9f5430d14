// oracle/ref_pieces_loss.cpp -- TEST INFRASTRUCTURE.  C-ABI harness around the REFERENCE'S OWN per-element loss header
// (/root/reference/inst/include/FactorNet/math/loss.hpp + core/constants.hpp: plain C++, no Eigen), compiled from where
// it lies.  Output: oracle/_ref/libref_loss.so.  Pins the oracle's NB IRLS weight and NB negative log-likelihood
//  against the reference itself.
#include <FactorNet/math/loss.hpp>

extern "C" {
__attribute__((visibility("default"))) double ref_irls_weight_nb_f64(double predicted, double nb_size) {
    return FactorNet::irls_weight_nb<double>(predicted, nb_size);
}
__attribute__((visibility("default"))) float ref_irls_weight_nb_f32(float predicted, float nb_size) {
    return FactorNet::irls_weight_nb<float>(predicted, nb_size);
}
__attribute__((visibility("default"))) double ref_loss_nb_f64(double observed, double predicted, double nb_size) {
    return FactorNet::loss_contribution_nb<double>(observed, predicted, nb_size);
}
__attribute__((visibility("default"))) float ref_loss_nb_f32(float observed, float predicted, float nb_size) {
    return FactorNet::loss_contribution_nb<float>(observed, predicted, nb_size);
}
__attribute__((visibility("default"))) double ref_irls_weight_kl_f64(double predicted) { return FactorNet::irls_weight_kl<double>(predicted); }
__attribute__((visibility("default"))) float ref_irls_weight_kl_f32(float predicted) { return FactorNet::irls_weight_kl<float>(predicted); }
__attribute__((visibility("default"))) double ref_irls_weight_gp_f64(double observed, double predicted, double theta, double blend) {
    return FactorNet::irls_weight_gp<double>(observed, predicted, theta, blend);
}
__attribute__((visibility("default"))) float ref_irls_weight_gp_f32(float observed, float predicted, float theta, float blend) {
    return FactorNet::irls_weight_gp<float>(observed, predicted, theta, blend);
}
__attribute__((visibility("default"))) double ref_loss_gp_f64(double observed, double predicted, double theta) {
    return FactorNet::loss_contribution_gp<double>(observed, predicted, theta);
}
__attribute__((visibility("default"))) float ref_loss_gp_f32(float observed, float predicted, float theta) {
    return FactorNet::loss_contribution_gp<float>(observed, predicted, theta);
}
__attribute__((visibility("default"))) double ref_irls_weight_power_f64(double predicted, double power) { return FactorNet::irls_weight_power<double>(predicted, power); }
__attribute__((visibility("default"))) float ref_irls_weight_power_f32(float predicted, float power) { return FactorNet::irls_weight_power<float>(predicted, power); }
// deviance terms: loss_type 6 = Gamma, 7 = inverse Gaussian, 8 = Tweedie(power)
__attribute__((visibility("default"))) double ref_loss_dev_f64(int loss_type, double y, double mu, double power) {
    return loss_type == 6 ? FactorNet::loss_contribution_gamma<double>(y, mu)
         : loss_type == 7 ? FactorNet::loss_contribution_invgauss<double>(y, mu)
                          : FactorNet::loss_contribution_tweedie<double>(y, mu, power);
}
__attribute__((visibility("default"))) float ref_loss_dev_f32(int loss_type, float y, float mu, float power) {
    return loss_type == 6 ? FactorNet::loss_contribution_gamma<float>(y, mu)
         : loss_type == 7 ? FactorNet::loss_contribution_invgauss<float>(y, mu)
                          : FactorNet::loss_contribution_tweedie<float>(y, mu, power);
}
__attribute__((visibility("default"))) double ref_robust_modifier_f64(double r, double delta) { return FactorNet::robust_huber_modifier<double>(r, delta); }
__attribute__((visibility("default"))) float ref_robust_modifier_f32(float r, float delta) { return FactorNet::robust_huber_modifier<float>(r, delta); }
// compute_robust_loss through the reference's own LossConfig (type = LossType value, robust_delta, power_param)
__attribute__((visibility("default"))) double ref_robust_loss_f64(int loss_type, double y, double mu, double theta, double power, double delta) {
    FactorNet::LossConfig<double> c;
    c.type = static_cast<FactorNet::LossType>(loss_type); c.robust_delta = delta; c.power_param = power;
    return FactorNet::compute_robust_loss<double>(y, mu, c, theta);
}
__attribute__((visibility("default"))) float ref_robust_loss_f32(int loss_type, float y, float mu, float theta, float power, float delta) {
    FactorNet::LossConfig<float> c;
    c.type = static_cast<FactorNet::LossType>(loss_type); c.robust_delta = delta; c.power_param = power;
    return FactorNet::compute_robust_loss<float>(y, mu, c, theta);
}
__attribute__((visibility("default"))) double ref_loss_mse_f64(double observed, double predicted) {
    return FactorNet::loss_contribution_mse<double>(observed, predicted);
}
}

// core/constants.hpp (self-contained: <limits>, <type_traits> only), the reference's numerical constants and algorithm defaults on the
// NMF path, in the order tests/golden/make_ref_constants.py names them
extern "C" __attribute__((visibility("default"))) int ref_constants(double* out, int cap) {
    const double v[] = {FactorNet::tiny_num<double>(), (double)FactorNet::tiny_num<float>(), FactorNet::kl_epsilon<double>(),
                        FactorNet::CD_TOL, (double)FactorNet::CD_MAXIT, FactorNet::CD_ABS_TOL, FactorNet::NMF_TOL,
                        (double)FactorNet::NMF_MAXIT, (double)FactorNet::NMF_PATIENCE, FactorNet::DEFAULT_L1, FactorNet::DEFAULT_L2,
                        FactorNet::DEFAULT_L21, FactorNet::DEFAULT_GRAPH_LAMBDA, FactorNet::DEFAULT_HUBER_DELTA};
    const int n = (int)(sizeof(v) / sizeof(v[0]));
    for (int i = 0; i < n && i < cap; ++i) out[i] = v[i];
    return n;
}
