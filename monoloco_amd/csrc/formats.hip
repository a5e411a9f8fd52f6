// On-disk formats either side of the hot path (host code only; nothing here touches the GPU):
//   * OpenPifPaf `*.predictions.json` text  ->  boxes (m,5) + keypoints (m,3,17), i.e. json.load +
//     preprocess_pifpaf (reference monoloco/network/process.py:155-218) without Python lists;
//   * the KITTI result txt lines of save_txts (reference monoloco/eval/generate_kitti.py:202-253).
// All arithmetic is IEEE double in the order the reference's Python expressions evaluate, so the
// results are bit-identical to the Python path (tests/test_formats.py).
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/monoloco_hip.h"

namespace {

thread_local char g_ferr[256] = "";

int ffail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_ferr, sizeof(g_ferr), fmt, ap);
    va_end(ap);
    return code;
}

// ------------------------------------------------------------------ a small JSON reader
struct Cur {
    const char* p;
    const char* e;
    const char* b;
    const char* what = nullptr;  // first syntax error
    bool fail(const char* w) {
        if (!what) what = w;
        return false;
    }
    void ws() {
        while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
    }
    bool lit(const char* s) {
        const size_t n = strlen(s);
        if ((size_t)(e - p) >= n && memcmp(p, s, n) == 0) {
            p += n;
            return true;
        }
        return false;
    }
};

// string token; [s, s+n) is the raw content between the quotes (escapes left as they are)
bool read_string(Cur& c, const char*& s, size_t& n) {
    if (c.p >= c.e || *c.p != '"') return c.fail("expected a string");
    ++c.p;
    s = c.p;
    while (c.p < c.e && *c.p != '"') {
        if (*c.p == '\\') {
            ++c.p;
            if (c.p >= c.e) return c.fail("unterminated string");
        }
        ++c.p;
    }
    if (c.p >= c.e) return c.fail("unterminated string");
    n = (size_t)(c.p - s);
    ++c.p;
    return true;
}

// number token as Python's json module reads it (ints and floats alike end up as the nearest double;
// NaN / Infinity / -Infinity are accepted like json.loads does)
bool read_number(Cur& c, double& v) {
    if (c.lit("NaN")) {
        v = NAN;
        return true;
    }
    if (c.lit("Infinity")) {
        v = INFINITY;
        return true;
    }
    if (c.lit("-Infinity")) {
        v = -INFINITY;
        return true;
    }
    char buf[512];
    size_t n = 0;
    const char* q = c.p;
    while (q < c.e && n + 1 < sizeof(buf) &&
           ((*q >= '0' && *q <= '9') || *q == '-' || *q == '+' || *q == '.' || *q == 'e' || *q == 'E'))
        buf[n++] = *q++;
    if (n == 0) return c.fail("expected a number");
    buf[n] = 0;
    char* end = nullptr;
    v = strtod(buf, &end);
    if (end != buf + n) return c.fail("malformed number");
    c.p = q;
    return true;
}

bool skip_value(Cur& c, int depth);

bool skip_container(Cur& c, char close, bool object, int depth) {
    if (depth > 256) return c.fail("nesting too deep");
    ++c.p;
    c.ws();
    if (c.p < c.e && *c.p == close) {
        ++c.p;
        return true;
    }
    while (true) {
        c.ws();
        if (object) {
            const char* s;
            size_t n;
            if (!read_string(c, s, n)) return false;
            c.ws();
            if (c.p >= c.e || *c.p != ':') return c.fail("expected ':'");
            ++c.p;
        }
        if (!skip_value(c, depth + 1)) return false;
        c.ws();
        if (c.p < c.e && *c.p == ',') {
            ++c.p;
            continue;
        }
        if (c.p < c.e && *c.p == close) {
            ++c.p;
            return true;
        }
        return c.fail("expected ',' or a closing bracket");
    }
}

bool skip_value(Cur& c, int depth) {
    c.ws();
    if (c.p >= c.e) return c.fail("unexpected end of input");
    const char ch = *c.p;
    if (ch == '{') return skip_container(c, '}', true, depth);
    if (ch == '[') return skip_container(c, ']', false, depth);
    if (ch == '"') {
        const char* s;
        size_t n;
        return read_string(c, s, n);
    }
    if (c.lit("true") || c.lit("false") || c.lit("null")) return true;
    double v;
    return read_number(c, v);
}

// flat array of numbers; reads at most cap values, counts all
bool read_number_array(Cur& c, double* out, int cap, int& count) {
    c.ws();
    if (c.p >= c.e || *c.p != '[') return c.fail("expected an array of numbers");
    ++c.p;
    count = 0;
    c.ws();
    if (c.p < c.e && *c.p == ']') {
        ++c.p;
        return true;
    }
    while (true) {
        c.ws();
        double v;
        if (!read_number(c, v)) return false;
        if (count < cap) out[count] = v;
        ++count;
        c.ws();
        if (c.p < c.e && *c.p == ',') {
            ++c.p;
            continue;
        }
        if (c.p < c.e && *c.p == ']') {
            ++c.p;
            return true;
        }
        return c.fail("expected ',' or ']'");
    }
}

bool key_is(const char* s, size_t n, const char* k) { return strlen(k) == n && memcmp(s, k, n) == 0; }

struct Ann {
    double kps[51];
    double box[4];
    double score;
    bool has_kps, has_box, has_score;
};

// one annotation object; later duplicates of a key win, as in a Python dict
bool read_annotation(Cur& c, Ann& a, int64_t index, int& err_code) {
    a.has_kps = a.has_box = a.has_score = false;
    c.ws();
    if (c.p >= c.e || *c.p != '{') return c.fail("expected an annotation object");
    ++c.p;
    c.ws();
    if (c.p < c.e && *c.p == '}') {
        ++c.p;
        return true;
    }
    while (true) {
        c.ws();
        const char* s;
        size_t n;
        if (!read_string(c, s, n)) return false;
        c.ws();
        if (c.p >= c.e || *c.p != ':') return c.fail("expected ':'");
        ++c.p;
        c.ws();
        if (key_is(s, n, "keypoints")) {
            int cnt;
            if (!read_number_array(c, a.kps, 51, cnt)) return false;
            if (cnt != 51) {
                err_code = ML_ERR_SHAPE;
                ffail(ML_ERR_SHAPE, "annotation %lld: 'keypoints' holds %d numbers, expected 51 (17 x (x, y, c))",
                      (long long)index, cnt);
                return false;
            }
            a.has_kps = true;
        } else if (key_is(s, n, "bbox")) {
            int cnt;
            if (!read_number_array(c, a.box, 4, cnt)) return false;
            if (cnt != 4) {
                err_code = ML_ERR_SHAPE;
                ffail(ML_ERR_SHAPE, "annotation %lld: 'bbox' holds %d numbers, expected 4", (long long)index, cnt);
                return false;
            }
            a.has_box = true;
        } else if (key_is(s, n, "score")) {
            if (!read_number(c, a.score)) return false;
            a.has_score = true;
        } else if (!skip_value(c, 1)) {
            return false;
        }
        c.ws();
        if (c.p < c.e && *c.p == ',') {
            ++c.p;
            continue;
        }
        if (c.p < c.e && *c.p == '}') {
            ++c.p;
            return true;
        }
        return c.fail("expected ',' or '}'");
    }
}

// numpy's float64 mean of 17 values: pairwise_sum with 8 running partials, then the tail, then / n
double numpy_mean17(const double* a, int stride) {
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j * stride];
    for (int j = 0; j < 8; ++j) r[j] += a[(8 + j) * stride];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    res += a[16 * stride];
    return res / 17.0;
}

int parse_impl(const char* json, int64_t len, int has_im_size, double im_w, double im_h, int enlarge_boxes,
               double min_conf, int64_t cap, double* boxes, double* keypoints, int64_t* m_out, int64_t* n_ann) {
    if (!json || len < 0) return ffail(ML_ERR_ARG, "null JSON buffer");
    Cur c{json, json + len, json};
    c.ws();
    if (c.p >= c.e || *c.p != '[') return ffail(ML_ERR_ARG, "predictions JSON must be a top-level array of annotations");
    ++c.p;
    int64_t n = 0, m = 0;
    const double shrink = enlarge_boxes ? 1.0 : 2.0;
    c.ws();
    bool done = (c.p < c.e && *c.p == ']');
    if (done) ++c.p;
    while (!done) {
        Ann a;
        int err_code = 0;
        if (!read_annotation(c, a, n, err_code)) {
            if (err_code) return err_code;
            return ffail(ML_ERR_ARG, "JSON syntax error at byte %lld: %s", (long long)(c.p - c.b), c.what ? c.what : "?");
        }
        if (!a.has_kps) return ffail(ML_ERR_ARG, "annotation %lld has no 'keypoints'", (long long)n);
        if (!a.has_box) return ffail(ML_ERR_ARG, "annotation %lld has no 'bbox'", (long long)n);
        // process.py:168-201
        double box[4] = {a.box[0], a.box[1], a.box[2], a.box[3]};
        double conf, dh, dw;
        if (a.has_score) {  // bbox is x, y, w, h
            conf = a.score;
            dh = box[3] / (10.0 * shrink);
            dw = box[2] / (5.0 * shrink);
            box[2] += box[0];
            box[3] += box[1];
        } else {  // bbox is corners; confidence = mean keypoint confidence
            conf = numpy_mean17(a.kps + 2, 3);
            dh = (box[3] - box[1]) / (7.0 * shrink);
            dw = (box[2] - box[0]) / (3.5 * shrink);
            if (!(dh > -5.0 && dw > -5.0)) return ffail(ML_ERR_ARG, "annotation %lld: Bounding box <=0", (long long)n);
        }
        box[0] -= dw;
        box[1] -= dh;
        box[2] += dw;
        box[3] += dh;
        if (has_im_size) {
            if (!(box[0] > 0.0)) box[0] = 0.0;  // max(0, x) of Python: x only if x > 0
            if (!(box[1] > 0.0)) box[1] = 0.0;
            if (im_w < box[2]) box[2] = im_w;
            if (im_h < box[3]) box[3] = im_h;
        }
        if (conf >= min_conf) {
            if (boxes && keypoints) {
                if (m >= cap) return ffail(ML_ERR_ARG, "output capacity %lld too small", (long long)cap);
                double* bo = boxes + m * 5;
                for (int k = 0; k < 4; ++k) bo[k] = box[k];
                bo[4] = conf;
                double* ko = keypoints + m * 51;
                for (int j = 0; j < 17; ++j) {
                    ko[j] = a.kps[3 * j];
                    ko[17 + j] = a.kps[3 * j + 1];
                    ko[34 + j] = a.kps[3 * j + 2];
                }
            }
            ++m;
        }
        ++n;
        c.ws();
        if (c.p < c.e && *c.p == ',') {
            ++c.p;
            continue;
        }
        if (c.p < c.e && *c.p == ']') {
            ++c.p;
            break;
        }
        return ffail(ML_ERR_ARG, "JSON syntax error at byte %lld: expected ',' or ']'", (long long)(c.p - c.b));
    }
    c.ws();
    if (c.p != c.e) return ffail(ML_ERR_ARG, "JSON syntax error at byte %lld: trailing data", (long long)(c.p - c.b));
    if (m_out) *m_out = m;
    if (n_ann) *n_ann = n;
    return ML_OK;
}

// '%f' of Python: like printf, but a NaN never carries a sign
int put_f(char* out, int64_t cap, int64_t& pos, double v) {
    char buf[400];
    int n;
    if (std::isnan(v))
        n = snprintf(buf, sizeof(buf), "nan ");
    else
        n = snprintf(buf, sizeof(buf), "%f ", v);
    if (n < 0 || n >= (int)sizeof(buf)) return -1;
    if (out) {
        if (pos + n > cap) return -1;
        memcpy(out + pos, buf, (size_t)n);
    }
    pos += n;
    return 0;
}

int put_s(char* out, int64_t cap, int64_t& pos, const char* s) {
    const int64_t n = (int64_t)strlen(s);
    if (out) {
        if (pos + n > cap) return -1;
        memcpy(out + pos, s, (size_t)n);
    }
    pos += n;
    return 0;
}

}  // namespace

extern "C" {

const char* ml_formats_last_error(void) { return g_ferr; }

int ml_pifpaf_count(const char* json, int64_t len, int64_t* n_annotations) {
    if (!n_annotations) return ffail(ML_ERR_ARG, "null output pointer");
    return parse_impl(json, len, 0, 0., 0., 1, -INFINITY, 0, nullptr, nullptr, nullptr, n_annotations);
}

int ml_pifpaf_parse(const char* json, int64_t len, int has_im_size, double im_w, double im_h, int enlarge_boxes,
                    double min_conf, int64_t cap, double* boxes, double* keypoints, int64_t* m) {
    if (!boxes || !keypoints || !m) return ffail(ML_ERR_ARG, "null output pointer");
    return parse_impl(json, len, has_im_size, im_w, im_h, enlarge_boxes, min_conf, cap, boxes, keypoints, m, nullptr);
}

int ml_kitti_txt_format(int64_t m, const double* boxes, const double* xyz, const double* bi, const double* epi,
                        const double* alpha, const double* ry, const double* hwl, const double* zz_override,
                        const double* tt, const double* cat, double conf_scale, char* out, int64_t cap,
                        int64_t* written) {
    if (m < 0 || !written || (m > 0 && (!boxes || !xyz || !bi || !epi || !cat)))
        return ffail(ML_ERR_ARG, "ml_kitti_txt_format: bad argument");
    int64_t pos = 0;
    for (int64_t i = 0; i < m; ++i) {
        const double* box = boxes + i * 5;
        // generate_kitti.py:224-243, evaluated in the same order
        const double xx = xyz[i * 3 + 0] - (tt ? tt[0] : 0.0);
        const double yy = xyz[i * 3 + 1] - (tt ? tt[1] : 0.0);
        double zz = xyz[i * 3 + 2] - (tt ? tt[2] : 0.0);
        if (zz_override) zz = zz_override[i];
        const double a = alpha ? alpha[i] : -10.0;
        const double r = ry ? ry[i] : -10.0;
        const double conf = conf_scale * box[4] / (bi[i] / std::sqrt(xx * xx + yy * yy + zz * zz));
        int rc = put_s(out, cap, pos, cat[i] < 0.1 ? "Pedestrian " : "Cyclist ");
        rc |= put_s(out, cap, pos, "-1 -1 ");
        rc |= put_f(out, cap, pos, a);
        for (int k = 0; k < 4; ++k) rc |= put_f(out, cap, pos, box[k]);
        for (int k = 0; k < 3; ++k) rc |= put_f(out, cap, pos, hwl ? hwl[i * 3 + k] : 0.0);
        rc |= put_f(out, cap, pos, xx);
        rc |= put_f(out, cap, pos, yy);
        rc |= put_f(out, cap, pos, zz);
        rc |= put_f(out, cap, pos, r);
        rc |= put_f(out, cap, pos, conf);
        rc |= put_f(out, cap, pos, bi[i]);
        rc |= put_f(out, cap, pos, epi[i]);
        rc |= put_s(out, cap, pos, "\n");
        if (rc) return ffail(ML_ERR_ARG, "ml_kitti_txt_format: output buffer of %lld bytes too small", (long long)cap);
    }
    *written = pos;
    return ML_OK;
}

}  // extern "C"
