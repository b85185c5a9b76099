// swarm/integrator.hpp -- swarm::integrator and gpu::integrator, mirror of
// swarm/integrator.{hpp,cpp}.  The GPU subclass drives libswarm_hip.so (include/swarm_hip.h).
#pragma once
#include <memory>

#include "ensemble.hpp"
#include "log.hpp"
#include "plugin.hpp"
#include "swarm_hip.h"

namespace swarm {

class integrator;
typedef std::shared_ptr<integrator> Pintegrator;

// swarm/integrator.hpp:49-149
class integrator {
protected:
    defaultEnsemble _ens;
    double _destination_time;
    log::Pmanager _logman;
    int _max_iterations;
    int _max_attempts;
    virtual void launch_integrator() = 0;

public:
    const static int _default_max_iterations;
    const static int _default_max_attempts;
    integrator(const config &cfg);
    virtual ~integrator() {}
    // integrator.cpp:91-100
    virtual void integrate();
    virtual void flush_log() { _logman->flush(); }
    virtual defaultEnsemble &get_ensemble() { return _ens; }
    // the integrator shares the caller's host buffers: results appear in `ens` after integrate()
    virtual void set_ensemble(defaultEnsemble &ens) { _ens = ens; }
    virtual void set_destination_time(const double &t) { _destination_time = t; }
    virtual double get_destination_time() const { return _destination_time; }
    virtual int get_max_iterations() const { return _max_iterations; }
    virtual void set_max_iterations(const int &mi) { _max_iterations = mi; }
    virtual int get_max_attempts() const { return _max_attempts; }
    virtual void set_max_attempts(const int &ma) { _max_attempts = ma; }
    virtual void set_log_manager(log::Pmanager &l) { _logman = l; }
    // integrator.cpp:129-145
    static Pintegrator create(const config &cfg);
};

// integrator.cpp:61-89
int number_of_active_systems(defaultEnsemble ens);
int number_of_disabled_systems(defaultEnsemble ens);
void activate_inactive_systems(defaultEnsemble &ens);

namespace gpu {

// swarm/integrator.hpp:178-275 -- keeps a device mirror of the ensemble
class integrator : public swarm::integrator {
protected:
    swarm_hip_ensemble_t _dens; // device wave-chunk image, owned
    swarm_hip_ensemble_t _view; // a run of wave chunks of _dens (core_integrate with log_batches)
    const swarm_hip_ensemble_t *_launch_ens; // what launch_integrator() works on: &_dens, or &_view
    int _log_batches; // launches per attempt with a two-log manager (`log_batches`, not a reference key)
    void *_stream;
    int _active_count; // ACTIVE systems as counted after the last launch of the attempt loop; 0 = not known
    void release_device();

public:
    integrator(const config &cfg);
    virtual ~integrator();
    // integrator.cpp:102-117
    virtual void integrate();
    // the device-resident part: attempt loop without upload/download
    virtual int core_integrate();
    virtual void set_ensemble(defaultEnsemble &ens);
    void upload_ensemble();
    void download_ensemble();
    const swarm_hip_ensemble_t &device_ensemble() const { return _dens; }
};

} // namespace gpu
} // namespace swarm
