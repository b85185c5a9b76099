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

manager::manager() : nbuf(1), cur(0), capacity(0), _dropped(0), copy_stream(0), ev_launch(0), cursor_image(0) {
    for (int i = 0; i < 2; i++) {
        d_buf[i] = 0, d_cursor[i] = 0, host[i] = 0, ev_copy[i] = 0, dirty[i] = false;
        pending[i].in_flight = false;
    }
}
manager::~manager() {
    if (copy_stream) {
        swarm_hip_synchronize(copy_stream);
        swarm_hip_stream_destroy(copy_stream);
    }
    if (ev_launch) swarm_hip_event_destroy(ev_launch);
    if (cursor_image) swarm_hip_host_free(cursor_image);
    for (int i = 0; i < 2; i++) {
        if (ev_copy[i]) swarm_hip_event_destroy(ev_copy[i]);
        if (d_buf[i]) swarm_hip_free(d_buf[i]);
        if (d_cursor[i]) swarm_hip_free(d_cursor[i]);
        if (host[i]) swarm_hip_host_free(host[i]);
    }
}
void manager::init(const config &cfg, int, int device_buffer_size) {
    log_writer.reset(writer::create(cfg));
    if (dynamic_cast<null_writer *>(log_writer.get())) return; // null sink: no device buffer
    capacity = device_buffer_size;
    nbuf = cfg.optional("log_buffers", 1) == 2 ? 2 : 1; // not a reference key
    for (int i = 0; i < nbuf; i++) {
        hipErrCheck(swarm_hip_malloc((void **)&d_buf[i], (size_t)capacity));
        hipErrCheck(swarm_hip_malloc((void **)&d_cursor[i], 16));
        hipErrCheck(swarm_hip_host_alloc((void **)&host[i], (size_t)capacity));
        hipErrCheck(swarm_hip_memset(d_cursor[i], 0, 16, 0));
    }
    if (nbuf == 2) {
        hipErrCheck(swarm_hip_stream_create(&copy_stream));
        hipErrCheck(swarm_hip_event_create(&ev_launch));
        for (int i = 0; i < 2; i++) hipErrCheck(swarm_hip_event_create(&ev_copy[i]));
        hipErrCheck(swarm_hip_host_alloc((void **)&cursor_image, 16));
    }
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

// what manager::flush does with the host copy (logmanager.cpp:79-86)
void manager::process(int i, int32_t used, int32_t dropped, int32_t nprintf) {
    _dropped += dropped;
    if (used > 0) {
        if (nprintf) replay_printf(std::cerr, (const char *)host[i], (size_t)used); // logmanager.cpp:83
        log_writer->process((const char *)host[i], (size_t)used);
    }
}
void manager::finish(int i) {
    if (!pending[i].in_flight) return;
    hipErrCheck(swarm_hip_event_synchronize(ev_copy[i]));
    pending[i].in_flight = false;
    process(i, pending[i].used, pending[i].dropped, pending[i].nprintf);
}
void manager::flush() {
    if (!d_buf[0]) return;
    for (int i = 0; i < nbuf; i++) {
        finish(i);
        if (dirty[i]) {
            hipErrCheck(swarm_hip_memset(d_cursor[i], 0, 16, 0));
            dirty[i] = false;
        }
    }
    int32_t c[4] = {0, 0, 0, 0}; // bytes used, dropped, lprintf records, -
    hipErrCheck(swarm_hip_memcpy_d2h(c, d_cursor[cur], 16));
    if (c[0] > 0) hipErrCheck(swarm_hip_memcpy_d2h(host[cur], d_buf[cur], (size_t)c[0]));
    process(cur, c[0], c[1], c[2]);
    if (c[0] || c[1]) hipErrCheck(swarm_hip_memset(d_cursor[cur], 0, 16, 0));
    hipErrCheck(swarm_hip_synchronize(0));
}
void manager::flush_async(void *launch_stream) {
    if (!overlapping()) {
        if (launch_stream) hipErrCheck(swarm_hip_synchronize(launch_stream));
        flush();
        return;
    }
    const int i = cur;
    // the cursor travels on the LAUNCH stream, right behind the kernels: the host learns it the moment they finish,
    // while the previous log's copy is still in flight on the copy stream, and queues this log's copy behind it
    hipErrCheck(swarm_hip_memcpy_d2h_async(cursor_image, d_cursor[i], 16, launch_stream));
    hipErrCheck(swarm_hip_event_record(ev_launch, launch_stream));
    hipErrCheck(swarm_hip_event_synchronize(ev_launch));
    pending_t p = {true, cursor_image[0], cursor_image[1], cursor_image[2]};
    if (p.used > 0) hipErrCheck(swarm_hip_memcpy_d2h_async(host[i], d_buf[i], (size_t)p.used, copy_stream));
    hipErrCheck(swarm_hip_event_record(ev_copy[i], copy_stream));
    pending[i] = p;
    dirty[i] = p.used || p.dropped;
    cur = 1 - i;
    finish(cur);
    if (dirty[cur]) { // cleared on the launch stream, ahead of the launch that appends to it next
        hipErrCheck(swarm_hip_memset(d_cursor[cur], 0, 16, launch_stream));
        dirty[cur] = false;
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
// integrator.cpp:129-145: the plugin "integrator_<name>" builds the object; an unknown name is a configuration error
Pintegrator integrator::create(const config &cfg) {
    const config::const_iterator chosen = cfg.find("integrator");
    if (chosen == cfg.end()) ERROR("Integrator type has to be chosen with 'integrator=xxx' keyword");
    plugins_map_t::const_iterator entry = plugins_map().find("integrator_" + chosen->second);
    if (entry == plugins_map().end() || !entry->second) ERROR("Integrator " + chosen->second + " not found.");
    return Pintegrator(static_cast<integrator *>(entry->second->create(cfg)));
}

// ---- gpu::integrator -------------------------------------------------------------------------
namespace gpu {

integrator::integrator(const config &cfg) : swarm::integrator(cfg), _stream(0), _active_count(0) {
    std::memset(&_dens, 0, sizeof(_dens));
    _view = _dens;
    _launch_ens = &_dens;
    _log_batches = std::max(1, cfg.optional("log_batches", 1));
}
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
    _active_count = 0;
    // `log_batches` with a two-log manager: every attempt as that many launches over consecutive runs of wave chunks
    // (a run of chunks is a contiguous range of each device buffer), the flush of one batch's log overlapping the next
    // batch's launch.  Every system is still launched once per attempt: same records, same counts.
    const bool overlap = _logman->overlapping();
    const long long nchunk = swarm_hip_nchunk(_dens.nsys);
    const int nb = (int)std::max(1LL, std::min((long long)(overlap ? _log_batches : 1), nchunk));
    for (int i = 0; i < _max_attempts; i++) {
        for (int k = 0; k < nb; k++) {
            if (nb > 1) {
                const long long c0 = nchunk * k / nb, c1 = nchunk * (k + 1) / nb;
                _view = _dens;
                _view.nsys = (int32_t)(std::min((long long)_dens.nsys, c1 * SWARM_HIP_CHUNK) - c0 * SWARM_HIP_CHUNK);
                _view.body = _dens.body + c0 * _dens.nbod * (7 + _dens.nbattr) * SWARM_HIP_CHUNK;
                _view.sysd = _dens.sysd + c0 * (1 + _dens.nsattr) * SWARM_HIP_CHUNK;
                _view.sysi = _dens.sysi + c0 * 2 * SWARM_HIP_CHUNK;
                _launch_ens = &_view;
                _active_count = 0;
            }
            launch_integrator(); // hands _active_count on as swarm_hip_launch_t::active_hint
            _launch_ens = &_dens;
            if (overlap)
                _logman->flush_async(_stream);
            else
                _logman->flush();
        }
        attempts++;
        int32_t n = 0;
        hipErrCheck(swarm_hip_count_active(&_dens, _stream, &n));
        if (n == 0) break;
        _active_count = n;
    }
    if (overlap) _logman->flush();
    _active_count = 0;
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
        // one wavefront or four per workgroup: two or three measured up to 2x slower at one wavefront per SIMD (swarm_hip.h)
        if (spb > 0) a.block_size = spb <= 64 ? 64 : 256;
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
        a.active_hint = _active_count;
        hipErrCheck(swarm_hip_launch_integrator(_launch_ens, &a));
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

// ---- snapshot: the two ensemble file formats of swarm/snapshot.{hpp,cpp} --------------------------------------------
// One description of what a file holds, read and written by two codecs.  A file is a header (nbod, nsys and the
// attribute counts), then per system a system record and nbod body records.  The formats differ in the order of the
// fields inside a record and in how a number is encoded:
//   binary (snapshot.hpp:153-170)  header {nsys, nbod, nsysattr, nbodattr} as int32; sys {time, id, state, attr[]};
//                                  body {pos[3], vel[3], mass, attr[]}; doubles and int32 as they are in memory
//   text   (snapshot.cpp:181-216)  "SwarmDataFile 2" / "nbod nsys nsysattr nbodattr" / "id time state" / attributes /
//                                  per body mass, position, velocity, attributes; every real as %.15le
// A file may carry fewer attributes than the library has (the rest stay 0), never more.
namespace snapshot {
namespace {

enum Field { F_ID, F_TIME, F_STATE, F_SYS_ATTR, F_MASS, F_POS, F_VEL, F_BODY_ATTR };
// one field of a record; `after` is what the text writer puts behind (each element of) it
struct Slot {
    Field field;
    const char *after;
};
const Slot BINARY_SYS[] = {{F_TIME, ""}, {F_ID, ""}, {F_STATE, ""}, {F_SYS_ATTR, ""}};
const Slot BINARY_BODY[] = {{F_POS, ""}, {F_VEL, ""}, {F_MASS, ""}, {F_BODY_ATTR, ""}};
const Slot TEXT_SYS[] = {{F_ID, " "}, {F_TIME, " "}, {F_STATE, "\n"}, {F_SYS_ATTR, " "}};
const Slot TEXT_BODY[] = {{F_MASS, "\n\t"}, {F_POS, " "}, {F_VEL, " "}, {F_BODY_ATTR, " "}};
const char *TEXT_TAG = "SwarmDataFile";
const int TEXT_VERSION = 2;

struct Counts {
    int nsys, nbod, nsysattr, nbodattr;
};

// where the elements of a field live in the host ensemble
struct Span {
    double *real;
    int *integer;
    int count;
};
Span locate(ensemble::Data &d, Field f, int sys, int bod, const Counts &c) {
    const size_t sb = (size_t)sys * d.nbod + bod;
    switch (f) {
    case F_ID: return Span{0, &d.id[sys], 1};
    case F_STATE: return Span{0, &d.state[sys], 1};
    case F_TIME: return Span{&d.time[sys], 0, 1};
    case F_SYS_ATTR: return Span{d.sattr.data() + (size_t)sys * ensemble::NUM_SYS_ATTRIBUTES, 0, c.nsysattr};
    case F_MASS: return Span{&d.mass[sb], 0, 1};
    case F_POS: return Span{&d.pos[sb * 3], 0, 3};
    case F_VEL: return Span{&d.vel[sb * 3], 0, 3};
    case F_BODY_ATTR: return Span{d.battr.data() + sb * ensemble::NUM_BODY_ATTRIBUTES, 0, c.nbodattr};
    }
    return Span{0, 0, 0};
}

// ---- codecs: how one number, and the punctuation around records, goes to or comes from the file
struct BinaryIn {
    FILE *f;
    const std::string &fn;
    template <class T> void number(T &v, const char *) {
        if (fread(&v, 1, sizeof(T), f) != sizeof(T)) throw readfileexception(fn, "File I/O error");
    }
    void text(const char *) {}
};
struct BinaryOut {
    FILE *f;
    const std::string &fn;
    template <class T> void number(T &v, const char *) {
        if (fwrite(&v, 1, sizeof(T), f) != sizeof(T)) throw writefileexception(fn, "File I/O error");
    }
    void text(const char *) {}
};
struct TextIn { // numbers separated by any white space, as fscanf takes them
    FILE *f;
    const std::string &fn;
    void number(double &v, const char *) {
        if (fscanf(f, "%le", &v) != 1) throw readfileexception(fn, "File I/O error");
    }
    void number(int &v, const char *) {
        if (fscanf(f, "%i", &v) != 1) throw readfileexception(fn, "File I/O error");
    }
    void text(const char *) {}
};
struct TextOut {
    FILE *f;
    const std::string &fn;
    void number(double &v, const char *after) { fprintf(f, "%.15le%s", v, after); }
    void number(int &v, const char *after) { fprintf(f, "%i%s", v, after); }
    void text(const char *t) { fputs(t, f); }
};

template <class Codec, size_t N>
void record(Codec &io, const Slot (&layout)[N], ensemble::Data &d, int sys, int bod, const Counts &c, const char *after_last_triple) {
    for (const Slot &slot : layout) {
        const Span s = locate(d, slot.field, sys, bod, c);
        for (int k = 0; k < s.count; k++) {
            // the last component of a position or velocity ends its line in the text format
            const bool ends_triple = (slot.field == F_POS || slot.field == F_VEL) && k == 2;
            const char *after = ends_triple ? after_last_triple : slot.after;
            if (s.real)
                io.number(s.real[k], after);
            else
                io.number(s.integer[k], after);
        }
    }
}

// every record of the file, in file order
template <class Codec> void all_records(Codec &io, bool text, ensemble::Data &d, const Counts &c) {
    for (int sys = 0; sys < c.nsys; sys++) {
        if (text) {
            record(io, TEXT_SYS, d, sys, 0, c, "");
            io.text("\n");
            for (int bod = 0; bod < c.nbod; bod++) {
                io.text("\t");
                record(io, TEXT_BODY, d, sys, bod, c, "\n\t");
                io.text("\n\n");
            }
            io.text("\n");
        } else {
            record(io, BINARY_SYS, d, sys, 0, c, "");
            for (int bod = 0; bod < c.nbod; bod++) record(io, BINARY_BODY, d, sys, bod, c, "");
        }
    }
}

struct File { // closes on every way out
    FILE *f;
    File(const std::string &fn, const char *mode) : f(fopen(fn.c_str(), mode)) {}
    ~File() {
        if (f) fclose(f);
    }
};

void check_attribute_counts(const Counts &c, const std::string &fn) {
    if (c.nsysattr > ensemble::NUM_SYS_ATTRIBUTES)
        throw readfileexception(fn, "The file requires more system attributes than the library can handle");
    if (c.nbodattr > ensemble::NUM_BODY_ATTRIBUTES)
        throw readfileexception(fn, "The file requires more planet attributes than the library can handle");
}

defaultEnsemble read_file(const std::string &fn, bool text) {
    File file(fn, text ? "r" : "rb");
    if (!file.f) throw readfileexception(fn, "cannot open file");
    Counts c;
    if (text) {
        char tag[1024];
        int version = 0;
        if (fscanf(file.f, "%1023s %i", tag, &version) != 2 || strcmp(TEXT_TAG, tag) != 0)
            throw readfileexception(fn, "Invalid file, header doesn't match");
        if (version != TEXT_VERSION) throw readfileexception(fn, "Incorrect version");
        if (fscanf(file.f, "%i %i %i %i", &c.nbod, &c.nsys, &c.nsysattr, &c.nbodattr) != 4)
            throw readfileexception(fn, "Invalid file, header doesn't match");
    } else {
        BinaryIn in{file.f, fn};
        in.number(c.nsys, ""), in.number(c.nbod, ""), in.number(c.nsysattr, ""), in.number(c.nbodattr, "");
    }
    check_attribute_counts(c, fn);
    if (c.nsys < 0 || c.nbod < 0 || c.nsysattr < 0 || c.nbodattr < 0) throw readfileexception(fn, "Invalid file, header doesn't match");
    hostEnsemble ens = hostEnsemble::create(c.nbod, c.nsys);
    if (text) {
        TextIn in{file.f, fn};
        all_records(in, true, ens.data(), c);
    } else {
        BinaryIn in{file.f, fn};
        all_records(in, false, ens.data(), c);
    }
    return ens;
}

void write_file(defaultEnsemble &ens, const std::string &fn, bool text) {
    File file(fn, text ? "w" : "wb");
    if (!file.f) throw writefileexception(fn, "cannot open file");
    Counts c = {ens.nsys(), ens.nbod(), ensemble::NUM_SYS_ATTRIBUTES, ensemble::NUM_BODY_ATTRIBUTES};
    if (text) {
        fprintf(file.f, "%s %i\n%i %i %i %i\n\n\n", TEXT_TAG, TEXT_VERSION, c.nbod, c.nsys, c.nsysattr, c.nbodattr);
        TextOut out{file.f, fn};
        all_records(out, true, ens.data(), c);
    } else {
        BinaryOut out{file.f, fn};
        out.number(c.nsys, ""), out.number(c.nbod, ""), out.number(c.nsysattr, ""), out.number(c.nbodattr, "");
        all_records(out, false, ens.data(), c);
    }
}

} // namespace

defaultEnsemble load(const std::string &filename) { return read_file(filename, false); }
defaultEnsemble load_text(const std::string &filename) { return read_file(filename, true); }
void save(defaultEnsemble &ens, const std::string &filename) { write_file(ens, filename, false); }
void save_text(defaultEnsemble &ens, const std::string &filename) { write_file(ens, filename, true); }

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

// Cartesian state about a centre of mass GM -> osculating elements (a, e, i, O, w, M), the conversion the reference's
// query -k prints (swarm/kepler.h, after Mercury's mco_x2el): the plane from the angular momentum, the conic from the
// orbital energy, then per conic the eccentricity, the true anomaly `nu` and the mean anomaly.  TINY = 1e-8 decides
// "in the reference plane", "circular" and "parabolic" as it does there.
namespace {
const double TINY = 1.e-8, TWO_PI = 2. * M_PI;
struct Vec3 {
    double x, y, z;
    double dot(const Vec3 &o) const { return x * o.x + y * o.y + z * o.z; }
    Vec3 cross(const Vec3 &o) const { return Vec3{y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x}; }
};
double into_turn(double ang) { return ang < 0. ? ang + TWO_PI : ang; }
struct InPlane {
    double e, nu, M; // eccentricity, true anomaly, mean anomaly
};
// bound orbit: eccentric anomaly from r = a (1 - e cos E), its quadrant from the sign of r.v
InPlane on_ellipse(double a, double h2_over_GM, double r, bool inbound, double u) {
    const double one_minus_p_over_a = 1. - h2_over_GM / a;
    if (!(one_minus_p_over_a > TINY)) return InPlane{0., u, u}; // circular: anomalies are counted from the node
    const double e = sqrt(one_minus_p_over_a);
    const double cosE = (a - r) / (a * e);
    double E = cosE > 1. ? 0. : (cosE > -1. ? acos(cosE) : M_PI);
    if (inbound) E = TWO_PI - E;
    const double denom = 1. - e * cos(E);
    const double nu = into_turn(atan2(sqrt(1. - e * e) * sin(E) / denom, (cos(E) - e) / denom));
    return InPlane{e, nu, E - e * sin(E)};
}
// unbound orbit: hyperbolic anomaly F from r = a (e cosh F - 1), a > 0 here
InPlane on_hyperbola(double a, double h2_over_GM, double r, bool inbound) {
    const double p_over_a = h2_over_GM / a;
    auto anomaly = [&](double e) {
        const double coshF = (a + r) / (a * e);
        return std::log(coshF + sqrt(coshF * coshF - 1.)); // acosh
    };
    if (!(p_over_a > TINY)) { // degenerate: a straight line through the centre, e = 1
        double nu = acos(h2_over_GM / r - 1.);
        if (inbound) nu = TWO_PI - nu;
        const double F = anomaly(1.);
        return InPlane{1., nu, sinh(F) - F};
    }
    const double e = sqrt(1. + p_over_a);
    double F = anomaly(e);
    if (inbound) F = -F;
    const double denom = e * cosh(F) - 1.;
    const double nu = into_turn(atan2(sqrt(e * e - 1.) * sinh(F) / denom, (e - cosh(F)) / denom));
    return InPlane{e, nu, e * sinh(F) - F};
}
} // namespace

void calc_keplerian_for_cartesian(double &a, double &e, double &i, double &O, double &w, double &M, const double x,
                                  const double y, const double z, const double vx, const double vy, const double vz,
                                  const double GM) {
    const Vec3 pos{x, y, z}, vel{vx, vy, vz};
    const Vec3 h = pos.cross(vel);
    const double h2 = h.dot(h), hmag = sqrt(h2);
    i = acos(h.z / hmag);
    // u = argument of latitude: the angle from the ascending node to the body
    double u;
    if (sqrt(h.x * h.x + h.y * h.y) / hmag < TINY) { // orbit in the reference plane: no node, count from +x
        O = 0.;
        u = atan2(y, x);
        if (fabs(i - M_PI) < 10. * TINY) u = -u;
    } else {
        O = atan2(h.x, -h.y);
        u = atan2(z / sin(i), x * cos(O) + y * sin(O));
    }
    O = into_turn(O);
    u = into_turn(u);
    const double r = sqrt(pos.dot(pos));
    const bool inbound = pos.dot(vel) < 0.;
    const double energy = vel.dot(vel) * 0.5 - GM / r;
    InPlane orbit;
    if (fabs(energy * r / GM) < sqrt(TINY)) {
        // parabola: a is the pericentre distance.  The reference forms the mean anomaly from its output argument w
        // before assigning it (kepler.h:133-135 reads `w` where it means the true anomaly); callers pass fresh
        // variables, which is the value 0 used here, so M = 0 outbound and tan(pi)(1 + ...) inbound -- kept, since
        // that is what the reference prints.
        a = 0.5 * h2 / GM;
        const double half_tan = tan(0.5 * (inbound ? TWO_PI : 0.)); // tan(pi) = -1.2e-16 in doubles, printed as such
        orbit = InPlane{1., acos(2. * a / r - 1.), half_tan * (1. + half_tan * half_tan / 3.)};
    } else if (energy < 0) {
        a = -0.5 * GM / energy;
        orbit = on_ellipse(a, h2 / GM, r, inbound, u);
    } else {
        a = 0.5 * GM / energy;
        orbit = on_hyperbola(a, h2 / GM, r, inbound);
    }
    e = orbit.e;
    M = orbit.M;
    w = into_turn(u - orbit.nu); // argument of pericentre, folded into (-pi, pi]
    w -= round(w / TWO_PI) * TWO_PI;
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
// Records with a negative msgid are gpulog's own (MSG_PRINTF = -2: lprintf lines) and carry no (T, sys) heading:
// their key is (-1, -1), so they sort ahead of every event and `query` lists them with T = -1 (get_Tsys, io.cpp:46-58).
struct rec_t {
    const char *ptr;
    int len;
    int msgid() const { int v; memcpy(&v, ptr, 4); return v; }
    bool has_heading() const { return msgid() >= 0 && len >= 20; }
    double T() const {
        if (!has_heading()) return -1;
        double v;
        memcpy(&v, ptr + 8, 8);
        return v;
    }
    int sys() const {
        if (!has_heading()) return -1;
        int v;
        memcpy(&v, ptr + 16, 4);
        return v;
    }
    bool operator<(const rec_t &a) const { // by time, then system, then position in the file
        const double Tt = T(), Ta = a.T();
        // (a NaN time, possible only in a damaged file, orders before every number: a strict weak ordering for std::sort)
        const bool nt = Tt != Tt, na = Ta != Ta;
        if (nt != na) return nt;
        if (!nt && Tt != Ta) return Tt < Ta;
        const int st = sys(), sa = a.sys();
        return st != sa ? st < sa : ptr < a.ptr;
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
        // a record is at least its own 8-byte header, a multiple of 8 long (gpulog pads to 16) and inside the file
        if (r.len < 8 || r.len % 8 != 0 || off + (size_t)r.len > data.size()) ERROR("corrupt log record");
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
    if (r.len < 32) ERROR("corrupt log record"); // before anything is read from it
    memcpy(&flags, r.ptr + 20, 4);
    memcpy(&nbod, r.ptr + 24, 4);
    if (nbod < 1 || 32 + (long long)sizeof(lbody) * nbod > r.len) ERROR("corrupt log record");
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
            if (r.len < 40) ERROR("corrupt log record");
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

// ---- utils: the helpers of swarm/utils.{hpp,cpp} --------------------------------------------------------------------
// generate_ensemble (utils.cpp:63-103).  In this fork the generator is deterministic (its random phase is commented
// out, L89-90), so every system is the same: a unit-mass star at rest at the origin and planets of mass planet_mass on
// circles (ejection_factor 1/sqrt 2) or faster orbits at r_k = 10 spacing_factor^(k-1), phase k pi/4.  One system is
// laid out and copied nsys times.
hostEnsemble generate_ensemble(config &cfg) {
    const int nsys = cfg.require("nsys", 0), nbod = cfg.require("nbod", 0);
    const double spacing = cfg.optional("spacing_factor", 1.4), planet_mass = cfg.optional("planet_mass", .0001);
    const double speed_factor = cfg.optional("ejection_factor", 1.0 / sqrt(2));
    const double star_radius = cfg.optional("star_radius", 0.1), planet_radius = cfg.optional("planet_radius", 0.001);
    const double star_mass = 1.;
    hostEnsemble ens = hostEnsemble::create(nbod, nsys);
    if (nsys <= 0 || nbod <= 0) return ens;
    ensemble::Data &d = ens.data();
    const size_t nb = nbod, na = ensemble::NUM_BODY_ATTRIBUTES;
    d.mass[0] = star_mass;
    d.battr[0] = star_radius;
    for (size_t k = 1; k < nb; k++) {
        const double r = 10 * pow(spacing, int(k - 1)), speed = sqrt(2 * star_mass / r) * speed_factor, phase = k * M_PI / 4;
        const double at[3] = {r * cos(phase), r * sin(phase), 0}, going[3] = {-speed * sin(phase), speed * cos(phase), 0.};
        for (int c = 0; c < 3; c++) d.pos[k * 3 + c] = at[c], d.vel[k * 3 + c] = going[c];
        d.mass[k] = planet_mass;
        d.battr[k * na] = planet_radius;
    }
    for (size_t sys = 1; sys < (size_t)nsys; sys++) {
        std::copy(d.pos.begin(), d.pos.begin() + nb * 3, d.pos.begin() + sys * nb * 3);
        std::copy(d.vel.begin(), d.vel.begin() + nb * 3, d.vel.begin() + sys * nb * 3);
        std::copy(d.mass.begin(), d.mass.begin() + nb, d.mass.begin() + sys * nb);
        std::copy(d.battr.begin(), d.battr.begin() + nb * na, d.battr.begin() + sys * nb * na);
    }
    for (int sys = 0; sys < nsys; sys++) d.id[sys] = sys; // time 0 and state ACTIVE (0) are create()'s zeros
    return ens;
}

// |dE/E| of every system between two snapshots of an ensemble, summarised (utils.cpp:108-121)
ensemble::range_t energy_conservation_error_range(ensemble &ens, ensemble &reference_ensemble) {
    std::vector<double> before(reference_ensemble.nsys()), after(ens.nsys());
    reference_ensemble.calc_total_energy(before.data());
    ens.calc_total_energy(after.data());
    for (size_t k = 0; k < after.size(); k++) after[k] = fabs((after[k] - before[k]) / before[k]);
    return ensemble::range_t::calculate(after.begin(), after.end());
}
double find_max_energy_conservation_error(ensemble &ens, ensemble &reference_ensemble) {
    return energy_conservation_error_range(ens, reference_ensemble).max;
}

// utils.cpp:124-138 accepts nsys in [1, 256000] and nbod in [3, MAX_NBODIES]; here the bounds are what the kernels are
// instantiated for.  A key that is absent counts as the reference's stand-in values (1000 systems, 3 bodies).
bool validate_configuration(config &cfg) {
    const int nsys = cfg.optional("nsys", 1000), nbod = cfg.optional("nbod", 3);
    return nsys >= 1 && nsys <= (1 << 24) && nbod >= SWARM_HIP_MIN_NBOD && nbod <= SWARM_HIP_MAX_NBOD;
}

// the "# key<TAB>value" block that heads a benchmark's CSV (utils.cpp:140-150)
void outputConfigSummary(std::ostream &o, config &cfg) {
    static const char *const rows[][2] = {{"Integrator:", "integrator"},          {"Time step", "time_step"},
                                          {"Destination time", "destination_time"}, {"Min time step", "min_time_step"},
                                          {"Max time step", "max_time_step"},       {"No. Systems", "nsys"},
                                          {"No. Bodies", "nbod"}};
    for (const auto &row : rows) o << "# " << row[0] << "\t" << cfg[row[1]] << "\n";
    o << std::endl;
}

// `swarm --defaults` (utils.cpp:154-162).  hermite_cpu is the reference's choice; the name is served by the HIP
// Hermite kernel here (see the plugin table above).
config default_config() {
    config cfg;
    static const char *const defaults[][2] = {
        {"integrator", "hermite_cpu"}, {"time_step", "0.001"}, {"log_writer", "null"}, {"nsys", "16"}, {"nbod", "3"}};
    for (const auto &kv : defaults) cfg[kv[0]] = kv[1];
    return cfg;
}

// "median[min-median,max-median] ", or the median alone when the spread is nil (utils.cpp:163-168)
std::ostream &operator<<(std::ostream &o, const ensemble::range_t &r) {
    if (r.min - r.max < 1e-11) return o << r.median;
    return o << r.median << "[" << (r.min - r.median) << "," << (r.max - r.median) << "] ";
}

// The three distances the tool reports (utils.cpp:170-214): the largest Euclidean separation of a body between the two
// ensembles, in position and in velocity, and the largest relative difference of the systems' times.  Plain
// reductions over the flat host arrays.
bool compare_ensembles(ensemble &e1, ensemble &e2, double &pos_diff, double &vel_diff, double &time_diff) {
    if (e1.nsys() != e2.nsys() || e1.nbod() != e2.nbod()) return false;
    pos_diff = vel_diff = time_diff = 0;
    if (e1.empty() || e2.empty()) return true;
    const ensemble::Data &a = e1.data(), &b = e2.data();
    auto largest_separation = [](const std::vector<double> &p, const std::vector<double> &q) {
        double worst = 0;
        for (size_t k = 0; k + 2 < p.size(); k += 3) {
            const double dx = p[k] - q[k], dy = p[k + 1] - q[k + 1], dz = p[k + 2] - q[k + 2];
            const double sep = sqrt(dx * dx + dy * dy + dz * dz);
            if (sep > worst) worst = sep;
        }
        return worst;
    };
    pos_diff = largest_separation(a.pos, b.pos);
    vel_diff = largest_separation(a.vel, b.vel);
    for (size_t k = 0; k < a.time.size(); k++) {
        const double rel = fabs(a.time[k] - b.time[k]) / (a.time[k] + b.time[k]) * 2;
        if (rel > time_diff) time_diff = rel;
    }
    return true;
}

} // namespace swarm

// C entry for the Python API (swarm_amd/swarmng.py: calc_keplerian_for_cartesian, module.cpp:65-73)
extern "C" void swarmng_calc_keplerian_for_cartesian(double *out6, double x, double y, double z, double vx, double vy,
                                                     double vz, double GM) {
    swarm::query::calc_keplerian_for_cartesian(out6[0], out6[1], out6[2], out6[3], out6[4], out6[5], x, y, z, vx, vy, vz, GM);
}
