s|            c6_wave_gamma<J>(need, gam, gscratch, lane, E.alpha, E.gam_d, E.gam_c, k0, k1, t, (uint32_t)d);|            for (int j_ = 0; j_ < J; j_++) noise_[j_] = 1.0f / (float)nchild;|
s|^            c6_wave_dirichlet<J>(gam, noise_);$|            (void)gam;|
