// swarm/ensemble.hpp -- host ensemble with the reference's accessor surface
// (swarm/types/ensemble.hpp: ens[i][b][c].pos(), sys.time(), state codes, calc_total_energy,
// time_ranges; swarm/ensemble_alloc.hpp:20-68: value type over shared buffers, create/clone).
// Host storage is plain AoS; the device image is the wave-chunk layout of include/swarm_hip.h.
#pragma once
#include <algorithm>
#include <cmath>
#include <memory>
#include <vector>

#include "common.hpp"

namespace swarm {

class ensemble {
public:
#ifndef SWARM_NUM_PLANET_ATTRIBUTES
#define SWARM_NUM_PLANET_ATTRIBUTES 1 // src/config.h.in NUM_PLANET_ATTRIBUTES, cmake/variables.cmake:47 default
#endif
    static const int NUM_BODY_ATTRIBUTES = SWARM_NUM_PLANET_ATTRIBUTES;
    static const int NUM_SYS_ATTRIBUTES = 1;
    // ensemble.hpp:269-274
    enum { SYSTEM_ACTIVE = 0, SYSTEM_INACTIVE = 1, SYSTEM_NEEDS_EXAM = 2, SYSTEM_DISABLED = -1 };

    struct Data {
        int nsys, nbod;
        std::vector<double> pos, vel, mass, time, battr, sattr;
        std::vector<int> state, id;
    };

    struct Component {
        double *_p, *_v;
        double &pos() const { return *_p; }
        double &vel() const { return *_v; }
    };
    struct BodyRef {
        Data *d;
        int sys, bod;
        Component operator[](int c) const {
            size_t k = ((size_t)sys * d->nbod + bod) * 3 + c;
            return Component{&d->pos[k], &d->vel[k]};
        }
        double &mass() const { return d->mass[(size_t)sys * d->nbod + bod]; }
        double &attribute(int l) const { return d->battr[((size_t)sys * d->nbod + bod) * NUM_BODY_ATTRIBUTES + l]; }
        double distance_to_origin_squared() const {
            const BodyRef &b = *this;
            return b[0].pos() * b[0].pos() + b[1].pos() * b[1].pos() + b[2].pos() * b[2].pos();
        }
        double speed_squared() const {
            const BodyRef &b = *this;
            return b[0].vel() * b[0].vel() + b[1].vel() * b[1].vel() + b[2].vel() * b[2].vel();
        }
        double distance_to_origin() const { return std::sqrt(distance_to_origin_squared()); }
        void get(double &x, double &y, double &z, double &vx, double &vy, double &vz) const {
            const BodyRef &b = *this;
            x = b[0].pos(), y = b[1].pos(), z = b[2].pos(), vx = b[0].vel(), vy = b[1].vel(), vz = b[2].vel();
        }
    };
    struct SystemRef {
        Data *d;
        int sys;
        BodyRef operator[](int b) const { return BodyRef{d, sys, b}; }
        int nbod() const { return d->nbod; }
        int number() const { return sys; }
        double &time() const { return d->time[sys]; }
        int &state() const { return d->state[sys]; }
        int &id() const { return d->id[sys]; }
        double &attribute(int l) const { return d->sattr[(size_t)sys * NUM_SYS_ATTRIBUTES + l]; }
        bool is_active() const { return state() == SYSTEM_ACTIVE; }
        bool is_inactive() const { return state() == SYSTEM_INACTIVE; }
        bool is_disabled() const { return state() == SYSTEM_DISABLED; }
        bool is_enabled() const { return !is_disabled(); }
        void set_active() const { state() = SYSTEM_ACTIVE; }
        void set_inactive() const { state() = SYSTEM_INACTIVE; }
        void set_disabled() const { state() = SYSTEM_DISABLED; }
        double distance_between(int i, int j) const {
            BodyRef a = (*this)[i], b = (*this)[j];
            double s = 0;
            for (int c = 0; c < 3; c++) s += (a[c].pos() - b[c].pos()) * (a[c].pos() - b[c].pos());
            return std::sqrt(s);
        }
        // ensemble.hpp:402-413
        double calc_total_energy() const {
            double K = 0.0, U = 0.0;
            for (int i = 0; i < nbod(); i++) K += 0.5 * (*this)[i].mass() * (*this)[i].speed_squared();
            for (int i = 0; i < nbod(); i++)
                for (int j = 0; j < i; j++) U += -(*this)[i].mass() * (*this)[j].mass() / distance_between(i, j);
            return K + U;
        }
    };

    // ensemble.hpp:705-726
    struct range_t {
        double average, min, max, median;
        range_t(double a, double m, double M, double d) : average(a), min(m), max(M), median(d) {}
        template <class It> static range_t calculate(It begin, It end) {
            size_t n = end - begin;
            double min = *begin, max = *begin, sum = *begin;
            for (It i = begin + 1; i != end; i++) {
                double v = *i;
                if (v < min) min = v;
                if (v > max) max = v;
                sum += v;
            }
            std::nth_element(begin, begin + n / 2, end);
            return range_t(sum / n, min, max, *(begin + n / 2));
        }
    };

    ensemble() {}
    static ensemble create(int nbod, int nsys) {
        ensemble e;
        e.d = std::make_shared<Data>();
        Data &D = *e.d;
        D.nsys = nsys, D.nbod = nbod;
        D.pos.assign((size_t)nsys * nbod * 3, 0.0);
        D.vel.assign((size_t)nsys * nbod * 3, 0.0);
        D.mass.assign((size_t)nsys * nbod, 0.0);
        D.time.assign(nsys, 0.0);
        D.battr.assign((size_t)nsys * nbod * NUM_BODY_ATTRIBUTES, 0.0);
        D.sattr.assign((size_t)nsys * NUM_SYS_ATTRIBUTES, 0.0);
        D.state.assign(nsys, 0);
        D.id.assign(nsys, 0);
        return e;
    }
    ensemble clone() const {
        ensemble e;
        e.d = std::make_shared<Data>(*d);
        return e;
    }
    bool empty() const { return !d; }
    int nsys() const { return d ? d->nsys : 0; }
    int nbod() const { return d ? d->nbod : 0; }
    SystemRef operator[](int i) const { return SystemRef{d.get(), i}; }
    Data &data() const { return *d; }

    void set_body(int sys, int bod, double m, double x, double y, double z, double vx, double vy, double vz) {
        BodyRef b = (*this)[sys][bod];
        b.mass() = m;
        b[0].pos() = x, b[1].pos() = y, b[2].pos() = z, b[0].vel() = vx, b[1].vel() = vy, b[2].vel() = vz;
    }
    void calc_total_energy(double *E) const {
        for (int s = 0; s != nsys(); s++) E[s] = (*this)[s].calc_total_energy();
    }
    range_t time_ranges() const {
        std::vector<double> times(d->time);
        return range_t::calculate(times.begin(), times.end());
    }

private:
    std::shared_ptr<Data> d;
};

typedef ensemble defaultEnsemble;
typedef ensemble hostEnsemble;

} // namespace swarm
