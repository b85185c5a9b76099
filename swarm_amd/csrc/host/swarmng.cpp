// swarmng.cpp -- implementation of the C++ host mirror (libswarmng_hip.so): plugin registry,
// integrators over the C-ABI, log manager + writers, snapshot I/O, utils.
// Reference counterparts are cited per function; nothing here computes trajectories on the CPU.
#include <cmath>
#include <cstring>
#include <limits>
#include <unistd.h>

#include <algorithm>
#include <fstream>
#include <regex>

#include "swarm/swarm.h"

int DEBUG_LEVEL = 0;

namespace swarm {

void hipException::check(int rc) {
    if (rc != 0) throw hipException(std::string("swarm_hip: ") + swarm_hip_last_error());
}

// ---- plugin registry: swarm/plugin.cpp:28-77 -------------------------------------------------
typedef std::map<std::string, plugin *> plugins_map_t;
static plugins_map_t &plugins_map() {
    static plugins_map_t pmap;
    return pmap;
}
void plugin::add(plugin *p) { plugins_map()[p->id()] = p; }
plugin *plugin::find(const std::string &name) {
    plugins_map_t::iterator i = plugins_map().find(name);
    if (i == plugins_map().end() || i->second == 0) throw plugin_not_found(name);
    return i->second;
}
void *plugin::instance(const std::string &name, const config &cfg) { return plugin::find(name)->create(cfg); }
std::vector<plugin *> plugin::all() {
    std::vector<plugin *> v;
    for (plugins_map_t::iterator i = plugins_map().begin(); i != plugins_map().end(); i++) v.push_back(i->second);
    return v;
}
plugin::help_t plugin::help;
std::ostream &operator<<(std::ostream &out, const plugin::help_t &) {
    std::vector<plugin *> ps = plugin::all();
    out << "Found " << ps.size() << " plugins " << std::endl;
    for (size_t i = 0; i < ps.size(); i++) out << ps[i]->id() << "\t" << ps[i]->description() << std::endl;
    return out;
}

// ---- log -------------------------------------------------------------------------------------
namespace log {

writer *writer::create(const config &cfg) {
    std::string name = cfg.optional("log_writer", std::string("null"));
    std::string plugin_name = "writer_" + name;
    try {
        return (writer *)plugin::instance(plugin_name, cfg);
    } catch (plugin_not_found &e) {
        ERROR("Log writer " + name + " not found.");
    }
}

binary_writer::binary_writer(const config &cfg) {
    binfn = cfg.optional("log_output", std::string());
    if (binfn == "") ERROR("Expected filename for writer.");
    rawfn = binfn + ".raw";
    f = fopen(rawfn.c_str(), "wb");
    if (!f) ERROR("Could not open '" + rawfn + "' for writing");
    swarm_header fh(query::UNSORTED_HEADER_FULL);
    fwrite(&fh, 1, sizeof(fh), f);
}
binary_writer::~binary_writer() {
    if (f) fclose(f);
    f = 0;
    try {
        if (query::sort_binary_log_file(binfn, rawfn)) unlink(rawfn.c_str());
    } catch (...) {
    }
}
void binary_writer::process(const char *p, size_t n) { if (n) fwrite(p, 1, n, f); }

static writer_plugin_initializer<null_writer> null_writer_plugin("null", "This is the dummy writer");
static writer_plugin_initializer<host_array_writer> host_array_writer_plugin("host_array", "Keeps the log in host memory");
static writer_plugin_initializer<binary_writer> binary_writer_plugin("binary", "This is the binary writer");

manager::~manager() {
    if (d_buf) swarm_hip_free(d_buf);
    if (d_cursor) swarm_hip_free(d_cursor);
    if (host) swarm_hip_host_free(host);
}
void manager::init(const config &cfg, int, int device_buffer_size) {
    log_writer.reset(writer::create(cfg));
    if (dynamic_cast<null_writer *>(log_writer.get())) return; // null sink: no device buffer
    capacity = device_buffer_size;
    hipErrCheck(swarm_hip_malloc((void **)&d_buf, (size_t)capacity));
    hipErrCheck(swarm_hip_malloc((void **)&d_cursor, 16));
    hipErrCheck(swarm_hip_host_alloc((void **)&host, (size_t)capacity));
    hipErrCheck(swarm_hip_memset(d_cursor, 0, 16, 0));
    hipErrCheck(swarm_hip_synchronize(0));
}
Pmanager &manager::default_log() {
    static Pmanager lm;
    if (!lm) {
        lm.reset(new manager());
        config cfg;
        cfg["log_writer"] = "null";
        lm->init(cfg, 0, 0);
    }
    return lm;
}
// run_printf / replay_printf, swarm/log/gpulog/lprintf.h:74-195: format every MSG_PRINTF record
// (header | char fmt[N] at 8 | arguments at natural alignment) and write it to `out`.
std::string run_printf(const char *rec) {
    int32_t msgid;
    std::memcpy(&msgid, rec, 4);
    if (msgid != -2 /* gpulog::MSG_PRINTF, bits/gpulog_constants.h:35 */) return "";
    std::string fmt(rec + 8);
    size_t at = 8 + fmt.size() + 1;
    std::string out;
    char buf[1024];
    size_t pos = 0;
    while (pos < fmt.size()) {
        size_t pc = fmt.find('%', pos);
        if (pc == std::string::npos) break;
        out.append(fmt, pos, pc - pos);
        size_t end = fmt.find_first_of("%cdiouxXeEfgGaAnps", pc + 1);
        if (end == std::string::npos) {
            pos = pc;
            break;
        }
        const std::string spec = fmt.substr(pc, end - pc + 1);
        const char k = fmt[end];
        if (std::strchr("cdiouxXp", k)) {
            at = (at + 3) / 4 * 4;
            int32_t v;
            std::memcpy(&v, rec + at, 4);
            at += 4;
            snprintf(buf, sizeof(buf), spec.c_str(), v);
            out += buf;
        } else if (std::strchr("eEfgGaA", k)) {
            at = (at + 7) / 8 * 8;
            double v;
            std::memcpy(&v, rec + at, 8);
            at += 8;
            snprintf(buf, sizeof(buf), spec.c_str(), v);
            out += buf;
        } else if (k == '%') {
            out += "%%"; // lprintf.h:154-156
        } else {
            out += spec;
        }
        pos = end + 1;
    }
    out.append(fmt, pos, std::string::npos);
    return out;
}
int replay_printf(std::ostream &out, const char *p, size_t n) {
    int count = 0;
    size_t off = 0;
    while (off + 8 <= n) {
        int32_t hdr[2];
        std::memcpy(hdr, p + off, 8);
        if (hdr[1] <= 0 || off + (size_t)hdr[1] > n) break;
        if (hdr[0] == -2) {
            out << run_printf(p + off);
            count++;
        }
        off += (size_t)hdr[1];
    }
    return count;
}

void manager::flush() {
    if (!d_buf) return;
    int32_t cur[4] = {0, 0, 0, 0}; // bytes used, dropped, lprintf records, -
    hipErrCheck(swarm_hip_memcpy_d2h(cur, d_cursor, 16));
    _dropped += cur[1];
    if (cur[0] > 0) {
        hipErrCheck(swarm_hip_memcpy_d2h(host, d_buf, (size_t)cur[0]));
        if (cur[2]) replay_printf(std::cerr, (const char *)host, (size_t)cur[0]); // logmanager.cpp:83
        log_writer->process((const char *)host, (size_t)cur[0]);
    }
    if (cur[0] || cur[1]) {
        hipErrCheck(swarm_hip_memset(d_cursor, 0, 16, 0));
        hipErrCheck(swarm_hip_synchronize(0));
    }
}

} // namespace log

// ---- integrator base: swarm/integrator.cpp ---------------------------------------------------
const int integrator::_default_max_iterations = 10000000;
const int integrator::_default_max_attempts = 1000000;

integrator::integrator(const config &cfg) : _destination_time(0) {
    _logman = log::manager::default_log();
    _max_iterations = cfg.optional("max_iterations", _default_max_iterations);
    _max_attempts = cfg.optional("max_attempts", _default_max_attempts);
}
int number_of_active_systems(defaultEnsemble ens) {
    int n = 0;
    for (int i = 0; i < ens.nsys(); i++)
        if (ens[i].is_active()) n++;
    return n;
}
int number_of_disabled_systems(defaultEnsemble ens) {
    int n = 0;
    for (int i = 0; i < ens.nsys(); i++)
        if (ens[i].is_disabled()) n++;
    return n;
}
void activate_inactive_systems(defaultEnsemble &ens) {
    for (int i = 0; i < ens.nsys(); i++)
        if (ens[i].is_inactive()) ens[i].set_active();
}
void integrator::integrate() {
    activate_inactive_systems(_ens);
    for (int i = 0; i < _max_attempts; i++) {
        launch_integrator();
        _logman->flush();
        if (number_of_active_systems(_ens) == 0) break;
    }
}
Pintegrator integrator::create(const config &cfg) {
    Pintegrator integ;
    if (!cfg.count("integrator")) ERROR("Integrator type has to be chosen with 'integrator=xxx' keyword");
    std::string name = cfg.at("integrator");
    std::string plugin_name = "integrator_" + name;
    try {
        integ.reset((integrator *)plugin::instance(plugin_name, cfg));
    } catch (plugin_not_found &e) {
        ERROR("Integrator " + name + " not found.");
    }
    return integ;
}

// ---- gpu::integrator -------------------------------------------------------------------------
namespace gpu {

integrator::integrator(const config &cfg) : swarm::integrator(cfg), _stream(0) { std::memset(&_dens, 0, sizeof(_dens)); }
integrator::~integrator() { release_device(); }
void integrator::release_device() {
    if (_dens.body) swarm_hip_free(_dens.body);
    if (_dens.sysd) swarm_hip_free(_dens.sysd);
    if (_dens.sysi) swarm_hip_free(_dens.sysi);
    std::memset(&_dens, 0, sizeof(_dens));
}
// integrator.hpp:233-236: clone the host ensemble's shape onto the device
void integrator::set_ensemble(defaultEnsemble &ens) {
    _ens = ens;
    release_device();
    if (ens.empty()) return; // an unset ensemble (swarm.cpp:219-220 sets the real one next)
    _dens.nsys = ens.nsys();
    _dens.nbod = ens.nbod();
    _dens.nbattr = ensemble::NUM_BODY_ATTRIBUTES;
    _dens.nsattr = ensemble::NUM_SYS_ATTRIBUTES;
    if (ens.nsys() == 0) return;
    hipErrCheck(swarm_hip_malloc((void **)&_dens.body, swarm_hip_body_bytes(_dens.nsys, _dens.nbod, _dens.nbattr)));
    hipErrCheck(swarm_hip_malloc((void **)&_dens.sysd, swarm_hip_sysd_bytes(_dens.nsys, _dens.nsattr)));
    hipErrCheck(swarm_hip_malloc((void **)&_dens.sysi, swarm_hip_sysi_bytes(_dens.nsys)));
}
void integrator::upload_ensemble() {
    ensemble::Data &D = _ens.data();
    hipErrCheck(swarm_hip_upload_aos(&_dens, D.pos.data(), D.vel.data(), D.mass.data(), D.time.data(), D.state.data(),
                                     D.id.data(), D.battr.data(), D.sattr.data()));
}
void integrator::download_ensemble() {
    ensemble::Data &D = _ens.data();
    hipErrCheck(swarm_hip_download_aos(&_dens, D.pos.data(), D.vel.data(), D.mass.data(), D.time.data(),
                                       D.state.data(), D.id.data(), D.battr.data(), D.sattr.data()));
}
int integrator::core_integrate() {
    int attempts = 0;
    for (int i = 0; i < _max_attempts; i++) {
        launch_integrator();
        _logman->flush();
        attempts++;
        int32_t n = 0;
        hipErrCheck(swarm_hip_count_active(&_dens, _stream, &n));
        if (n == 0) break;
    }
    return attempts;
}
void integrator::integrate() {
    activate_inactive_systems(_ens);
    if (_ens.nsys() == 0) return;
    upload_ensemble();
    core_integrate();
    download_ensemble();
}

} // namespace gpu

// ---- the plugins: one class, parameterised by kernel and monitor ------------------------------
// Constructor-time key handling follows each reference constructor (cited inline).
template <int ID, int MON, bool MCE_NAME = false> class hip_integrator : public gpu::integrator {
    swarm_hip_launch_t a;

public:
    hip_integrator(const config &cfg) : gpu::integrator(cfg) {
        std::memset(&a, 0, sizeof(a));
        a.integrator = ID;
        a.monitor = MON;
        int spb = cfg.optional("system_per_block", 0); // the reference's block-shape knob
        if (spb > 0) a.block_size = std::max(64, std::min(256, (spb + 63) / 64 * 64));
        if (ID == SWARM_HIP_HERMITE || ID == SWARM_HIP_MVS || ID == SWARM_HIP_EULER || ID == SWARM_HIP_MIDPOINT ||
            ID == SWARM_HIP_HERMITE_PROP || ID == SWARM_HIP_IRK2 || ID == SWARM_HIP_IRK2_CPU)
            a.time_step = cfg.require("time_step", 0.0); // hermite_integrator.hpp:44-46, mvs.hpp:41-43, ...
        if (ID == SWARM_HIP_HERMITE_ADAP) { // hermite_adap.hpp:44-47
            a.time_step_factor = cfg.require("time_step_factor", 0.0);
            a.min_time_step = cfg.require("min_time_step", 0.0);
        }
        if (ID == SWARM_HIP_RKCK_FIXED || ID == SWARM_HIP_RKCK_ADAPTIVE) { // rkck_integrator.hpp:63-71
            if (!cfg.count("min_time_step"))
                ERROR("Integrator rkck requires a min timestep ('min time step' keyword in the config file).");
            a.min_time_step = atof(cfg.at("min_time_step").c_str());
            if (!cfg.count("max_time_step"))
                ERROR("Integrator rkck requires a max timestep ('max time step' keyword in the config file).");
            a.max_time_step = atof(cfg.at("max_time_step").c_str());
            if (!cfg.count("error_tolerance"))
                ERROR("Integrator rkck requires a error tolerance ('error tolerance' keyword in the config file).");
            a.error_tolerance = atof(cfg.at("error_tolerance").c_str());
        }
        if (ID == SWARM_HIP_VERLET) { // verlet.hpp:35-39
            a.max_timestep = cfg.require("max_timestep", 0.0);
            a.timestep_scale = cfg.require("timestep_scale", 0.0);
        }
        if (MON == SWARM_HIP_MON_EJECTION || MON == SWARM_HIP_MON_EJECTION_OR_CLOSE) { // stop_on_ejection.hpp:47-53
            a.rmax = cfg.optional("rmax", (double)std::numeric_limits<float>::max());
            a.deactivate_on_ejection = cfg.optional("deactivate_on_ejection", false);
            a.log_on_ejection = cfg.optional("log_on_ejection", false);
            a.verbose_on_ejection = cfg.optional("verbose_on_ejection", false);
        }
        if (MON == SWARM_HIP_MON_EJECTION_OR_CLOSE) { // stop_on_close_encounter.hpp:43-49
            a.close_approach = cfg.optional("close_approach", 0.0);
            a.deactivate_on_close_encounter = cfg.optional("deactivate_on_close_encounter", false);
            a.log_on_close_encounter = cfg.optional("log_on_close_encounter", false);
            a.verbose_on_close_encounter = cfg.optional("verbose_on_close_encounter", false);
        }
        if (MON == SWARM_HIP_MON_LOG_TRANSIT) { // log_transit_params, log_transit.hpp:48-58
            cfg.require("nbod", 0);
            a.log_transit_tol = cfg.optional("log_transit_tol", 2.e-8);
            a.time_step = cfg.require("time_step", 0.0);
            a.log_transits = cfg.optional("log_transits", true);
            a.log_occultations = cfg.optional("log_occultations", false);
        }
        if (MON == SWARM_HIP_MON_LOG_INTERVAL) a.log_interval = cfg.require("log_interval", 0.0); // log_time_interval.hpp:41
        if (ID == SWARM_HIP_HERMITE || ID == SWARM_HIP_IRK2) a.rceh = cfg.optional("rceh", 1e-6); // mce_stat_params, mercury_mce_stat.hpp:47
        if (ID == SWARM_HIP_IRK2) a.irk2_method = cfg.optional("method", 6); // irk2_integrator.hpp:53
        if (ID == SWARM_HIP_MVS) a.kepler_compensated = cfg.optional("kepler_compensated", 0); // not a reference key
        a.lane_mode = cfg.optional("lane_mode", 0); // not a reference key: 0 auto, 1 lane per system, 2 lane per body
    }
    // = launch_templatized_integrator(this), swarm/gpu/helpers.hpp:216-238: ONE C-ABI call
    virtual void launch_integrator() {
        // `hermite` is hermite<mce_stat> in this fork (plugins/hermite.cu:40).  mce_stat needs body
        // attributes 1 and 2, which only exist when the library is built with
        // NUM_PLANET_ATTRIBUTES >= 3 (make NUM_PLANET_ATTRIBUTES=3); with the default of 1 the
        // reference's monitor indexes past the Body struct, and `hermite` keeps stop_on_ejection.
        if (MCE_NAME)
            a.monitor = _dens.nbattr >= 3 ? SWARM_HIP_MON_MCE_STAT : SWARM_HIP_MON_EJECTION;
        a.destination_time = _destination_time;
        a.max_iterations = _max_iterations;
        a.log = _logman->get_gpulog();
        a.stream = _stream;
        hipErrCheck(swarm_hip_launch_integrator(&_dens, &a));
    }
};

// registry names = src/plugins/*.cu.  `hermite` binds mce_stat (plugins/hermite.cu:40) when the
// build has >= 3 body attributes and stop_on_ejection otherwise (see launch_integrator above).
// hermite_log is commented out in the fork.
#define REG(var, ID, MON, name, desc) static integrator_plugin_initializer<hip_integrator<ID, MON> > var(name, desc)
static integrator_plugin_initializer<hip_integrator<SWARM_HIP_HERMITE, SWARM_HIP_MON_EJECTION, true> > p_hermite("hermite", "Hermite PEC2, fixed time step (MI355X)");
// The reference's CPU plugin names (plugins/hermite_cpu.cpp:36-56).  There is no CPU integrator in
// this library; the names are kept so that the reference's own configurations (test/sample/CPU.cfg,
// test/integrators/Hermite_CPU.cfg, test/bdb/bdb.sh, `swarm --defaults`) run unchanged.  They are
// served by the same HIP Hermite kernel, whose arithmetic follows hermite_cpu
// (integrators/hermite_cpu.hpp) and which reproduces its golden outputs.
REG(p_hermite_cpu, SWARM_HIP_HERMITE, SWARM_HIP_MON_EJECTION, "hermite_cpu", "Hermite PEC2 with hermite_cpu's arithmetic, on the MI355X");
REG(p_hermite_cpu_ce, SWARM_HIP_HERMITE, SWARM_HIP_MON_EJECTION_OR_CLOSE, "hermite_cpu_ejection_or_close_encounter", "hermite_cpu + stop_on_ejection_or_close_encounter");
REG(p_hermite_cpu_log, SWARM_HIP_HERMITE, SWARM_HIP_MON_LOG_INTERVAL, "hermite_cpu_log", "hermite_cpu + log_time_interval");
REG(p_hermite_adap, SWARM_HIP_HERMITE_ADAP, SWARM_HIP_MON_EJECTION, "hermite_adap", "Hermite PEC2, adaptive time step (MI355X)");
REG(p_hermite_adap_log, SWARM_HIP_HERMITE_ADAP, SWARM_HIP_MON_LOG_INTERVAL, "hermite_adap_log", "hermite_adap + log_time_interval");
REG(p_hermite_adap_ce, SWARM_HIP_HERMITE_ADAP, SWARM_HIP_MON_EJECTION_OR_CLOSE, "hermite_adap_close_encounter", "hermite_adap + stop_on_ejection_or_close_encounter");
REG(p_hermite_adap_transit, SWARM_HIP_HERMITE_ADAP, SWARM_HIP_MON_LOG_TRANSIT, "hermite_adap_transit", "hermite_adap + log_transit (experimental in the reference)");
REG(p_rkck_fixed, SWARM_HIP_RKCK_FIXED, SWARM_HIP_MON_EJECTION, "rkck_fixed", "Cash-Karp Runge-Kutta, fixed time step (MI355X)");
REG(p_rkck_fixed_ce, SWARM_HIP_RKCK_FIXED, SWARM_HIP_MON_EJECTION_OR_CLOSE, "rkck_fixed_close_encounter", "rkck_fixed + stop_on_ejection_or_close_encounter");
REG(p_rkck_adaptive, SWARM_HIP_RKCK_ADAPTIVE, SWARM_HIP_MON_EJECTION, "rkck_adaptive", "Cash-Karp Runge-Kutta, adaptive time step (MI355X)");
REG(p_rkck_adaptive_ce, SWARM_HIP_RKCK_ADAPTIVE, SWARM_HIP_MON_EJECTION_OR_CLOSE, "rkck_adaptive_close_encounter", "rkck_adaptive + stop_on_ejection_or_close_encounter");
REG(p_mvs, SWARM_HIP_MVS, SWARM_HIP_MON_EJECTION, "mvs", "This is the integrator based on mvs propagator");
REG(p_mvs_ce, SWARM_HIP_MVS, SWARM_HIP_MON_EJECTION_OR_CLOSE, "mvs_close_encounter", "This is the integrator based on mvs propagator, monitor stop_on_ejection_or_close_encounter");
REG(p_euler, SWARM_HIP_EULER, SWARM_HIP_MON_EJECTION, "euler", "This is the integrator based on euler propagator");
REG(p_verlet, SWARM_HIP_VERLET, SWARM_HIP_MON_EJECTION, "verlet", "This is the integrator based on verlet propagator");
REG(p_midpoint, SWARM_HIP_MIDPOINT, SWARM_HIP_MON_EJECTION, "midpoint", "This is the integrator based on midpoint propagator");
static integrator_plugin_initializer<hip_integrator<SWARM_HIP_IRK2, SWARM_HIP_MON_EJECTION, true> > p_irk2("irk2", "Implicit Runge-Kutta w/ Fixed Time step (MI355X)");
REG(p_irk2_cpu, SWARM_HIP_IRK2_CPU, SWARM_HIP_MON_EJECTION, "irk2_cpu", "Implicit Runge-Kutta with irk2_cpu's constants, on the MI355X");
REG(p_hermite_prop, SWARM_HIP_HERMITE_PROP, SWARM_HIP_MON_EJECTION, "hermite_prop", "Propagator version of Hermite integrator");

// ---- swarm::init -----------------------------------------------------------------------------
void init(const config &cfg) {
    // select_cuda_device(cfg.optional("CUDA_DEVICE", ...)), swarm.h:49-60; HIP_DEVICE is an alias
    int dev = cfg.optional("HIP_DEVICE", cfg.optional("CUDA_DEVICE", 0));
    hipErrCheck(swarm_hip_init(dev));
    log::manager::default_log()->init(cfg);
}

// ---- snapshot: swarm/snapshot.cpp ------------------------------------------------------------
namespace snapshot {

template <class T> static void readfromFILE(FILE *f, T &t, const std::string &fn) {
    if (fread(&t, 1, sizeof(t), f) != sizeof(t)) throw readfileexception(fn, "File I/O error");
}
template <class T> static void writetoFILE(FILE *f, const T &t, const std::string &fn) {
    if (fwrite(&t, 1, sizeof(t), f) != sizeof(t)) throw writefileexception(fn, "File I/O error");
}

defaultEnsemble load(const std::string &filename) {
    FILE *f = fopen(filename.c_str(), "rb");
    if (!f) throw readfileexception(filename, "cannot open file");
    header h; sys s; body b;
    readfromFILE(f, h, filename);
    if (h.nsysattr > ensemble::NUM_SYS_ATTRIBUTES)
        throw readfileexception(filename, "The file requires more system attributes than the library can handle");
    if (h.nbodattr > ensemble::NUM_BODY_ATTRIBUTES)
        throw readfileexception(filename, "The file requires more planet attributes than the library can handle");
    hostEnsemble ens = hostEnsemble::create(h.nbod, h.nsys);
    for (int i = 0; i < h.nsys; i++) {
        ensemble::SystemRef sr = ens[i];
        readfromFILE(f, s, filename);
        sr.id() = s.id, sr.time() = s.time, sr.state() = s.state;
        for (int l = 0; l < h.nsysattr; l++) sr.attribute(l) = s.attribute[l];
        for (int j = 0; j < h.nbod; j++) {
            readfromFILE(f, b, filename);
            sr[j].mass() = b.mass;
            for (int c = 0; c < 3; c++) sr[j][c].pos() = b.pos[c], sr[j][c].vel() = b.vel[c];
            for (int l = 0; l < h.nbodattr; l++) sr[j].attribute(l) = b.attribute[l];
        }
    }
    fclose(f);
    return ens;
}

static const char *DEFAULT_IO_TAG = "SwarmDataFile";
static const int CURRENT_IO_VERSION = 2;

defaultEnsemble load_text(const std::string &filename) {
    FILE *f = fopen(filename.c_str(), "r");
    if (!f) throw readfileexception(filename, "cannot open file");
    header h;
    char tag[1024];
    int version = 0;
    if (fscanf(f, "%1023s %i\n", tag, &version) != 2 || strcmp(DEFAULT_IO_TAG, tag) != 0)
        throw readfileexception(filename, "Invalid file, header doesn't match");
    if (version != CURRENT_IO_VERSION) throw readfileexception(filename, "Incorrect version");
    if (fscanf(f, "%i %i %i %i\n\n\n", &h.nbod, &h.nsys, &h.nsysattr, &h.nbodattr) != 4)
        throw readfileexception(filename, "Invalid file, header doesn't match");
    if (h.nsysattr > ensemble::NUM_SYS_ATTRIBUTES)
        throw readfileexception(filename, "The file requires more system attributes than the library can handle");
    if (h.nbodattr > ensemble::NUM_BODY_ATTRIBUTES)
        throw readfileexception(filename, "The file requires more planet attributes than the library can handle");
    hostEnsemble ens = hostEnsemble::create(h.nbod, h.nsys);
    int ok = 1;
    for (int i = 0; i < h.nsys && ok; i++) {
        ensemble::SystemRef sr = ens[i];
        ok &= fscanf(f, "%i %le %i\n", &sr.id(), &sr.time(), &sr.state()) == 3;
        for (int l = 0; l < h.nsysattr; l++) ok &= fscanf(f, "%le ", &sr.attribute(l)) == 1;
        for (int j = 0; j < h.nbod; j++) {
            ok &= fscanf(f, "\t%le\n\t%le %le %le\n\t%le %le %le\n\n", &sr[j].mass(), &sr[j][0].pos(), &sr[j][1].pos(),
                         &sr[j][2].pos(), &sr[j][0].vel(), &sr[j][1].vel(), &sr[j][2].vel()) == 7;
            for (int l = 0; l < h.nbodattr; l++) ok &= fscanf(f, "%le", &sr[j].attribute(l)) == 1;
        }
    }
    fclose(f);
    if (!ok) throw readfileexception(filename, "File I/O error");
    return ens;
}

void save(defaultEnsemble &ens, const std::string &filename) {
    FILE *f = fopen(filename.c_str(), "wb");
    if (!f) throw writefileexception(filename, "cannot open file");
    header h; sys s; body b;
    memset(&s, 0, sizeof(s));
    memset(&b, 0, sizeof(b));
    h.nsys = ens.nsys(), h.nbod = ens.nbod();
    h.nsysattr = ensemble::NUM_SYS_ATTRIBUTES, h.nbodattr = ensemble::NUM_BODY_ATTRIBUTES;
    writetoFILE(f, h, filename);
    for (int i = 0; i < h.nsys; i++) {
        ensemble::SystemRef sr = ens[i];
        s.id = sr.id(), s.time = sr.time(), s.state = sr.state();
        for (int l = 0; l < h.nsysattr; l++) s.attribute[l] = sr.attribute(l);
        writetoFILE(f, s, filename);
        for (int j = 0; j < h.nbod; j++) {
            for (int c = 0; c < 3; c++) b.pos[c] = sr[j][c].pos(), b.vel[c] = sr[j][c].vel();
            b.mass = sr[j].mass();
            for (int l = 0; l < h.nbodattr; l++) b.attribute[l] = sr[j].attribute(l);
            writetoFILE(f, b, filename);
        }
    }
    fclose(f);
}

void save_text(defaultEnsemble &ens, const std::string &filename) {
    FILE *f = fopen(filename.c_str(), "w");
    if (!f) throw writefileexception(filename, "cannot open file");
    fprintf(f, "%s %i\n", DEFAULT_IO_TAG, CURRENT_IO_VERSION);
    fprintf(f, "%i %i %i %i\n\n\n", ens.nbod(), ens.nsys(), ensemble::NUM_SYS_ATTRIBUTES, ensemble::NUM_BODY_ATTRIBUTES);
    for (int i = 0; i < ens.nsys(); i++) {
        ensemble::SystemRef sr = ens[i];
        fprintf(f, "%i %.15le %i\n", sr.id(), sr.time(), sr.state());
        for (int l = 0; l < ensemble::NUM_SYS_ATTRIBUTES; l++) fprintf(f, "%.15le ", sr.attribute(l));
        fprintf(f, "\n");
        for (int j = 0; j < ens.nbod(); j++) {
            fprintf(f, "\t%.15le\n\t%.15le %.15le %.15le\n\t%.15le %.15le %.15le\n\t", sr[j].mass(), sr[j][0].pos(),
                    sr[j][1].pos(), sr[j][2].pos(), sr[j][0].vel(), sr[j][1].vel(), sr[j][2].vel());
            for (int l = 0; l < ensemble::NUM_BODY_ATTRIBUTES; l++) fprintf(f, "%.15le ", sr[j].attribute(l));
            fprintf(f, "\n\n");
        }
        fprintf(f, "\n");
    }
    fclose(f);
}

} // namespace snapshot

// ---- binary log files and query: swarm/log/io.cpp, swarm/query.cpp ------------------------------
std::string swarm_header::type() const {
    const char *c = strstr(m_type, "//");
    if (!c) return trim(m_type);
    return trim(std::string(m_type, c - m_type));
}

namespace query {

const char *UNSORTED_HEADER_FULL = "unsorted_output // Unsorted output file";
const char *UNSORTED_HEADER_CHECK = "unsorted_output";
const char *SORTED_HEADER_FULL = "T_sorted_output // Output file sorted by time";
const char *SORTED_HEADER_CHECK = "T_sorted_output";

static planets_coordinate_system_t planets_coordinate_system = jacobi; // query.cpp:65
void set_coordinate_system(const planets_coordinate_system_t &c) { planets_coordinate_system = c; }
static bool keplerian_output = false;
void set_keplerian_output() { keplerian_output = true; }

void calc_keplerian_for_cartesian(double &a, double &e, double &i, double &O, double &w, double &M, const double x,
                                  const double y, const double z, const double vx, const double vy, const double vz,
                                  const double GM) {
    const double TINY = 1.e-8, TWOPI = 2. * M_PI;
    // orbital plane from the angular momentum; u = argument of latitude
    const double h[3] = {y * vz - z * vy, z * vx - x * vz, x * vy - y * vx};
    const double h2 = h[0] * h[0] + h[1] * h[1] + h[2] * h[2], hh = sqrt(h2);
    i = acos(h[2] / hh);
    double fac = sqrt(h[0] * h[0] + h[1] * h[1]) / hh, u;
    if (fac < TINY) { // orbit in the reference plane: the node is undefined
        O = 0.;
        u = atan2(y, x);
        if (fabs(i - M_PI) < 10. * TINY) u = -u;
    } else {
        O = atan2(h[0], -h[1]);
        u = atan2(z / sin(i), x * cos(O) + y * sin(O));
    }
    if (O < 0.) O += TWOPI;
    if (u < 0.) u += TWOPI;
    const double r = sqrt(x * x + y * y + z * z), rdotv = vx * x + vy * y + vz * z;
    const double energy = (vx * vx + vy * vy + vz * vz) * 0.5 - GM / r;
    auto wrap = [&](double ang) { // kepler.cpp:141-142
        if (ang < 0.) ang += TWOPI;
        return ang - round(ang / TWOPI) * TWOPI;
    };
    if (fabs(energy * r / GM) < sqrt(TINY)) { // parabola (kepler.cpp:131-143; its `w` is read before it is set: 0 here)
        a = 0.5 * h2 / GM;
        e = 1.;
        const double ww = acos(2. * a / r - 1.);
        w = 0.;
        if (rdotv < 0.) w = TWOPI - w;
        const double tmpf = tan(0.5 * w);
        M = tmpf * (1. + tmpf * tmpf / 3.);
        w = wrap(u - ww);
    } else if (energy < 0) { // ellipse
        a = -0.5 * GM / energy;
        fac = 1. - h2 / (GM * a);
        double ww, cape;
        if (fac > TINY) {
            e = sqrt(fac);
            const double face = (a - r) / (a * e);
            cape = face > 1. ? 0. : (face > -1. ? acos(face) : M_PI);
            if (rdotv < 0.) cape = TWOPI - cape;
            const double cw = (cos(cape) - e) / (1. - e * cos(cape));
            const double sw = sqrt(1. - e * e) * sin(cape) / (1. - e * cos(cape));
            ww = atan2(sw, cw);
            if (ww < 0.) ww += TWOPI;
        } else {
            e = 0.;
            ww = u;
            cape = u;
        }
        M = cape - e * sin(cape);
        w = wrap(u - ww);
    } else { // hyperbola (energy > 0; energy == 0 is caught by the parabola test)
        a = 0.5 * GM / energy;
        fac = h2 / (GM * a);
        double ww, capf;
        if (fac > TINY) {
            e = sqrt(1. + fac);
            const double tmpf = (a + r) / (a * e);
            capf = std::log(tmpf + sqrt(tmpf * tmpf - 1.));
            if (rdotv < 0.) capf = -capf;
            const double cw = (e - cosh(capf)) / (e * cosh(capf) - 1.);
            const double sw = sqrt(e * e - 1.) * sinh(capf) / (e * cosh(capf) - 1.);
            ww = atan2(sw, cw);
            if (ww < 0.) ww += TWOPI;
        } else {
            e = 1.;
            double tmpf = 0.5 * h2 / GM;
            ww = acos(2. * tmpf / r - 1.);
            if (rdotv < 0.) ww = TWOPI - ww;
            tmpf = (a + r) / (a * e);
            capf = std::log(tmpf + sqrt(tmpf * tmpf - 1.));
        }
        M = e * sinh(capf) - capf;
        w = wrap(u - ww);
    }
}

template <typename T> static T arg_parse(const std::string &s) {
    if (s == "MIN") return range<T>().first;
    if (s == "MAX") return range<T>().last;
    return (T)atof(s.c_str());
}
template <typename T> range<T> range<T>::parse(const std::string &s) {
    if (s == "ALL") return range<T>();
    static std::regex r("(.+?)(?:\\.\\.(.+))?");
    std::smatch match;
    if (!std::regex_match(s, match, r)) ERROR("invalid value '" + s + "' for a range");
    if (match[2] == "") return range<T>(arg_parse<T>(match[1]));
    return range<T>(arg_parse<T>(match[1]), arg_parse<T>(match[2]));
}
template struct range<int>;
template struct range<double>;

// one record of the gpulog stream: header {int msgid; int len}, then (double T, int sys, ...)
struct rec_t {
    const char *ptr;
    int len;
    int msgid() const { int v; memcpy(&v, ptr, 4); return v; }
    double T() const { double v; memcpy(&v, ptr + 8, 8); return v; }
    int sys() const { int v; memcpy(&v, ptr + 16, 4); return v; }
    bool operator<(const rec_t &a) const { // idx_t::operator<, io.cpp:72-84
        double Tt = T(), Ta = a.T();
        int syst = sys(), sysa = a.sys();
        return Tt < Ta || (Tt == Ta && (syst < sysa || (syst == sysa && ptr < a.ptr)));
    }
};

static std::vector<char> read_log_file(const std::string &fn, std::string &type) {
    std::ifstream in(fn.c_str(), std::ios::binary);
    if (!in) ERROR("Could not open '" + fn + "' for reading");
    std::vector<char> all((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    if (all.size() < sizeof(swarm_header) || memcmp(all.data(), "SWARM\0" "0\0", 8) != 0)
        ERROR("'" + fn + "' is not a swarm log file");
    swarm_header fh("");
    memcpy((void *)&fh, all.data(), sizeof(fh));
    type = fh.type();
    return std::vector<char>(all.begin() + sizeof(swarm_header), all.end());
}

static std::vector<rec_t> index_records(const std::vector<char> &data) {
    std::vector<rec_t> idx;
    size_t off = 0;
    while (off + 8 <= data.size()) {
        rec_t r;
        r.ptr = data.data() + off;
        memcpy(&r.len, r.ptr + 4, 4);
        if (r.len <= 0 || off + (size_t)r.len > data.size()) ERROR("corrupt log record");
        idx.push_back(r);
        off += (size_t)r.len;
    }
    return idx;
}

bool sort_binary_log_file(const std::string &outfn, const std::string &infn) {
    std::string type;
    std::vector<char> data = read_log_file(infn, type);
    if (type != UNSORTED_HEADER_CHECK) ERROR("'" + infn + "' is not an unsorted log file");
    std::vector<rec_t> idx = index_records(data);
    std::sort(idx.begin(), idx.end());
    std::ofstream out(outfn.c_str(), std::ios::binary);
    if (!out) ERROR("Could not open '" + outfn + "' for writing");
    swarm_header fh(SORTED_HEADER_FULL, 0, data.size());
    out.write((const char *)&fh, sizeof(fh));
    for (size_t i = 0; i != idx.size(); i++) out.write(idx[i].ptr, idx[i].len);
    return true;
}

// log::body, log/types.hpp:37-56 (64 bytes)
struct lbody { double x, y, z, vx, vy, vz, mass; int body_id; int pad; };

static lbody center_of_mass(const lbody *bodies, const int nbod) { // query.cpp:95-119
    lbody c;
    memset(&c, 0, sizeof(c));
    for (int i = 0; i < nbod; i++) {
        c.x += bodies[i].x * bodies[i].mass, c.y += bodies[i].y * bodies[i].mass, c.z += bodies[i].z * bodies[i].mass;
        c.vx += bodies[i].vx * bodies[i].mass, c.vy += bodies[i].vy * bodies[i].mass, c.vz += bodies[i].vz * bodies[i].mass;
        c.mass += bodies[i].mass;
    }
    c.x /= c.mass, c.y /= c.mass, c.z /= c.mass, c.vx /= c.mass, c.vy /= c.mass, c.vz /= c.mass;
    return c;
}

// record_output_1 (EVT_SNAPSHOT), query.cpp:122-205, Cartesian branch
static void record_output_1(std::ostream &out, const rec_t &r, const body_range_t &body_range) {
    double time = r.T();
    int sys = r.sys(), flags, nbod;
    memcpy(&flags, r.ptr + 20, 4);
    memcpy(&nbod, r.ptr + 24, 4);
    std::vector<lbody> bodies(nbod);
    memcpy((void *)bodies.data(), r.ptr + 32, sizeof(lbody) * nbod);
    lbody center;
    const lbody &star = bodies[0];
    switch (planets_coordinate_system) {
    case astrocentric: center = star; break;
    case barycentric: center = center_of_mass(bodies.data(), nbod); break;
    case jacobi: center = star; break;
    case origin:
        memset(&center, 0, sizeof(center));
        center.mass = star.mass;
        break;
    }
    char buf[1000];
    for (int bod = 0; bod < nbod; bod++) {
        if (!body_range.in(bod)) continue;
        const lbody &b = bodies[bod];
        if ((planets_coordinate_system == jacobi) && (bod == 0)) continue;
        if (((planets_coordinate_system != jacobi) && (bod > 0)) || ((planets_coordinate_system == jacobi) && (bod > 1)))
            out << "\n";
        if (planets_coordinate_system == jacobi) center = center_of_mass(bodies.data(), bod);
        double x = b.x - center.x, y = b.y - center.y, z = b.z - center.z;
        double vx = b.vx - center.vx, vy = b.vy - center.vy, vz = b.vz - center.vz;
        if (keplerian_output && bod > 0) { // query.cpp:187-194
            double cm = center.mass;
            if (planets_coordinate_system == barycentric) cm -= b.mass;
            double ka, ke, ki, kO, kw, kM;
            calc_keplerian_for_cartesian(ka, ke, ki, kO, kw, kM, x, y, z, vx, vy, vz, b.mass + cm);
            const double rad2deg = 180. / M_PI;
            snprintf(buf, sizeof(buf), "%10d %lg  %6d %6d  %lg  % 9.5lg % 9.5lg % 9.5lg  % 9.5lg % 9.5lg % 9.5lg  %d", r.msgid(),
                     time, sys, bod, b.mass, ka, ke, ki * rad2deg, kO * rad2deg, kw * rad2deg, kM * rad2deg, flags);
            out << buf;
        }
        if (!keplerian_output) {
            snprintf(buf, sizeof(buf), "%10d %lg  %6d %6d  %lg  %9.5lg %9.5lg %9.5lg  %9.5lg %9.5lg %9.5lg  %d", r.msgid(), time,
                     sys, bod, b.mass, x, y, z, vx, vy, vz, flags);
            out << buf;
        }
    }
}

void execute(const std::string &datafile, time_range_t T, sys_range_t sys, body_range_t bod, std::ostream &out) {
    if (datafile.size() >= 3 && datafile.substr(datafile.size() - 3) == ".db")
        ERROR("Berkeley DB log files are not supported by this build");
    std::string type;
    std::vector<char> data = read_log_file(datafile, type);
    std::vector<rec_t> idx = index_records(data);
    if (type != SORTED_HEADER_CHECK) std::sort(idx.begin(), idx.end()); // a raw file: order it first
    for (size_t i = 0; i != idx.size(); i++) {
        const rec_t &r = idx[i];
        if (!T.in(r.T()) || !sys.in(r.sys())) continue;
        if (r.msgid() == 1)
            record_output_1(out, r, bod);
        else if (r.msgid() == 15 || r.msgid() == 16) { // record_output_15 / _16, query.cpp:243-294
            int body_id;
            double minb, vproj;
            memcpy(&body_id, r.ptr + 20, 4), memcpy(&minb, r.ptr + 24, 8), memcpy(&vproj, r.ptr + 32, 8);
            if (bod.in(body_id)) {
                char buf[1000];
                snprintf(buf, sizeof(buf), "%10d %lg  %6d %6d  %lg %lg %s Experimental", r.msgid(), r.T(), r.sys(), body_id, minb,
                         vproj, r.msgid() == 15 ? "Transit" : "Occultation");
                out << buf;
            }
        } else { // record_output_default
            char buf[200];
            snprintf(buf, sizeof(buf), "%10d %lg  %6d", r.msgid(), r.T(), r.sys());
            out << buf;
        }
        out << "\n";
    }
}

} // namespace query

// ---- utils: swarm/utils.cpp ------------------------------------------------------------------
hostEnsemble generate_ensemble(config &cfg) { // utils.cpp:63-103
    int nsys = cfg.require("nsys", 0);
    int nbod = cfg.require("nbod", 0);
    double spacing_factor = cfg.optional("spacing_factor", 1.4);
    double planet_mass = cfg.optional("planet_mass", .0001);
    double ejection_factor = cfg.optional("ejection_factor", 1.0 / sqrt(2));
    double star_radius = cfg.optional("star_radius", 0.1);
    double planet_radius = cfg.optional("planet_radius", 0.001);
    hostEnsemble ens = hostEnsemble::create(nbod, nsys);
    for (int sys = 0; sys < ens.nsys(); ++sys) {
        double mass_sun = 1.;
        double x = 0, y = 0, z = 0, vx = 0, vy = 0, vz = 0;
        ens.set_body(sys, 0, mass_sun, x, y, z, vx, vy, vz);
        ens[sys][0].attribute(0) = star_radius;
        for (int bod = 1; bod < ens.nbod(); ++bod) {
            double rmag = 10 * pow(spacing_factor, int(bod - 1));
            double vmag = sqrt(2 * mass_sun / rmag) * ejection_factor;
            double theta = bod * M_PI / 4; // deterministic in this fork (rand() commented out, L89-90)
            x = rmag * cos(theta), y = rmag * sin(theta), z = 0;
            vx = -vmag * sin(theta), vy = vmag * cos(theta), vz = 0.;
            ens.set_body(sys, bod, planet_mass, x, y, z, vx, vy, vz);
            ens[sys][bod].attribute(0) = planet_radius;
        }
        ens[sys].set_active();
        ens[sys].time() = 0;
        ens[sys].id() = sys;
    }
    return ens;
}

double find_max_energy_conservation_error(ensemble &ens, ensemble &reference_ensemble) {
    return energy_conservation_error_range(ens, reference_ensemble).max;
}
ensemble::range_t energy_conservation_error_range(ensemble &ens, ensemble &reference_ensemble) { // utils.cpp:108-121
    std::vector<double> energy_init(reference_ensemble.nsys()), energy_final(ens.nsys()), deltaE(ens.nsys());
    reference_ensemble.calc_total_energy(&energy_init[0]);
    ens.calc_total_energy(&energy_final[0]);
    for (int sysid = 0; sysid < ens.nsys(); ++sysid)
        deltaE[sysid] = fabs((energy_final[sysid] - energy_init[sysid]) / energy_init[sysid]);
    return ensemble::range_t::calculate(deltaE.begin(), deltaE.end());
}

// utils.cpp:124-138.  The reference rejects nsys > 256000 and nbod > MAX_NBODIES (6 by default);
// this build accepts what the kernels are instantiated for.
bool validate_configuration(config &cfg) {
    bool valid = true;
    int nsystems = cfg.optional("nsys", 1000);
    int nbodypersystem = cfg.optional("nbod", 3);
    if (!(nsystems >= 1) || !(nsystems <= (1 << 24))) valid = false;
    if (!(nbodypersystem >= SWARM_HIP_MIN_NBOD) || !(nbodypersystem <= SWARM_HIP_MAX_NBOD)) valid = false;
    return valid;
}

void outputConfigSummary(std::ostream &o, config &cfg) { // utils.cpp:140-150
    o << "# Integrator:\t" << cfg["integrator"] << "\n"
      << "# Time step\t" << cfg["time_step"] << "\n"
      << "# Destination time\t" << cfg["destination_time"] << "\n"
      << "# Min time step\t" << cfg["min_time_step"] << "\n"
      << "# Max time step\t" << cfg["max_time_step"] << "\n"
      << "# No. Systems\t" << cfg["nsys"] << "\n"
      << "# No. Bodies\t" << cfg["nbod"] << "\n"
      << std::endl;
}

// utils.cpp:154-162 sets integrator=hermite_cpu although docs/swarm_exec.dox:70-77 documents
// `hermite`; this build has no CPU integrators, so the documented value is used.
config default_config() {
    config cfg;
    cfg["integrator"] = "hermite_cpu"; // utils.cpp:157; served by the HIP Hermite kernel, see REG below
    cfg["time_step"] = "0.001";
    cfg["log_writer"] = "null";
    cfg["nsys"] = "16";
    cfg["nbod"] = "3";
    return cfg;
}

std::ostream &operator<<(std::ostream &o, const ensemble::range_t &r) { // utils.cpp:163-168
    if (r.min - r.max < 1e-11) return o << r.median;
    return o << r.median << "[" << (r.min - r.median) << "," << (r.max - r.median) << "] ";
}

static inline double square(double x) { return x * x; }
bool compare_ensembles(ensemble &e1, ensemble &e2, double &pos_diff, double &vel_diff, double &time_diff) { // utils.cpp:170-214
    if (e1.nsys() != e2.nsys() || e1.nbod() != e2.nbod()) return false;
    pos_diff = vel_diff = time_diff = 0;
    for (int i = 0; i < e1.nsys(); i++) {
        for (int j = 0; j < e1.nbod(); j++) {
            double dp = sqrt(square(e1[i][j][0].pos() - e2[i][j][0].pos()) + square(e1[i][j][1].pos() - e2[i][j][1].pos()) +
                             square(e1[i][j][2].pos() - e2[i][j][2].pos()));
            double dv = sqrt(square(e1[i][j][0].vel() - e2[i][j][0].vel()) + square(e1[i][j][1].vel() - e2[i][j][1].vel()) +
                             square(e1[i][j][2].vel() - e2[i][j][2].vel()));
            if (dp > pos_diff) pos_diff = dp;
            if (dv > vel_diff) vel_diff = dv;
        }
        double dt = fabs(e1[i].time() - e2[i].time()) / (e1[i].time() + e2[i].time()) * 2;
        if (dt > time_diff) time_diff = dt;
    }
    return true;
}

} // namespace swarm

// C entry for the Python API (swarm_amd/swarmng.py: calc_keplerian_for_cartesian, module.cpp:65-73)
extern "C" void swarmng_calc_keplerian_for_cartesian(double *out6, double x, double y, double z, double vx, double vy,
                                                     double vz, double GM) {
    swarm::query::calc_keplerian_for_cartesian(out6[0], out6[1], out6[2], out6[3], out6[4], out6[5], x, y, z, vx, vy, vz, GM);
}
