// swarm/utils.hpp -- generator, comparators, default config: mirror of swarm/utils.{hpp,cpp}.
#pragma once
#include <iostream>

#include "config.hpp"
#include "ensemble.hpp"

namespace swarm {

std::ostream &operator<<(std::ostream &o, const ensemble::range_t &r);
hostEnsemble generate_ensemble(config &cfg);
bool validate_configuration(config &cfg);
double find_max_energy_conservation_error(ensemble &ens, ensemble &reference_ensemble);
ensemble::range_t energy_conservation_error_range(ensemble &ens, ensemble &reference_ensemble);
bool compare_ensembles(ensemble &e1, ensemble &e2, double &pos_diff, double &vel_diff, double &time_diff);
config default_config();
void outputConfigSummary(std::ostream &o, config &cfg);

} // namespace swarm
