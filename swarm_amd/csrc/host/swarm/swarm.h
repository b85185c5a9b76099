// swarm/swarm.h -- public entry header, mirror of swarm/swarm.h.
#pragma once
#include "common.hpp"
#include "config.hpp"
#include "ensemble.hpp"
#include "integrator.hpp"
#include "log.hpp"
#include "plugin.hpp"
#include "query.hpp"
#include "snapshot.hpp"
#include "utils.hpp"

namespace swarm {
// swarm::init(cfg), swarm.h:49: select the device, initialise the default log manager
void init(const config &cfg);
}
