#!/usr/bin/env python3
"""Known-answer vectors for the one third-party formula on the path whose restatement was wrong by a rounding until round 5:
Distances.jl 0.10 `Haversine` (call sites src/gridcellgeometry.jl:187,221,246,261) converts the two LATITUDES to radians first and
subtracts after -- φ₁ = deg2rad(x[2]); φ₂ = deg2rad(y[2]); Δφ = φ₂ - φ₁ -- while the longitude difference is converted after the
subtraction, Δλ = deg2rad(y[1] - x[1]).  Rounds 1-4 computed Δφ = deg2rad(lat₂ - lat₁): the same number in exact arithmetic, another
double in the last place for most pairs.  The pairs below are chosen so that the two forms give DIFFERENT distances, and so that a
reader can redo the decisive step with nothing but IEEE double multiplication and subtraction (every value is printed as a hex
float): tests/test_gridmetrics.py demands the convert-first value from every copy of the formula (oracle C, Python transliteration,
host product, device kernel).  This script uses neither oracle/ nor the product.
    python tests/golden/known_answer/make_haversine_answer.py
"""
import json
import math
import os

HERE = os.path.dirname(os.path.abspath(__file__))
D2R = math.pi / 180  # deg2rad(z::Float64) = z * (pi / 180)
R = 6371000.0


def both(lon1, lat1, lon2, lat2):
    dl = (lon2 - lon1) * D2R
    p1, p2 = lat1 * D2R, lat2 * D2R
    out = {}
    for name, dp in (("convert_then_subtract", p2 - p1), ("subtract_then_convert", (lat2 - lat1) * D2R)):
        a = math.sin(dp / 2) ** 2 + math.cos(p1) * math.cos(p2) * math.sin(dl / 2) ** 2
        out[name] = {"dphi_hex": dp.hex(), "distance": 2 * (R * math.asin(min(math.sqrt(a), 1.0)))}
    out["phi1_hex"], out["phi2_hex"] = p1.hex(), p2.hex()
    return out


# the first three differ in the last place; in the last two the latitudes nearly cancel, φ₂ - φ₁ loses seven digits that deg2rad(lat₂ - lat₁)
# keeps, and the two forms differ by 1e-7 RELATIVE: a copy of the formula evaluated with another math library (numpy's, the device's: agreement
# to 1e-12 only) is still told apart
PAIRS = [(80.0, 0.3, 80.0, 0.7), (80.0, 10.1, 80.0, 10.4), (80.0, 0.3, 81.0, 33.9), (80.0, 64.7, 80.0, 64.7000001), (30.0, -12.3, 30.0, -12.29999999)]
doc = {"formula": "Distances.jl 0.10 Haversine: dlam = deg2rad(lon2 - lon1); phi1 = deg2rad(lat1); phi2 = deg2rad(lat2); dphi = phi2 - phi1; "
                  "a = sin(dphi/2)^2 + cos(phi1)*cos(phi2)*sin(dlam/2)^2; 2*(r*asin(min(sqrt(a), 1)))", "radius": R, "pairs": []}
for p in PAIRS:
    b = both(*p)
    assert b["convert_then_subtract"]["distance"] != b["subtract_then_convert"]["distance"], p
    b["relative_difference_of_the_two_forms"] = abs(b["convert_then_subtract"]["distance"] / b["subtract_then_convert"]["distance"] - 1)
    doc["pairs"].append({"lon1": p[0], "lat1": p[1], "lon2": p[2], "lat2": p[3], **b})
json.dump(doc, open(os.path.join(HERE, "haversine.json"), "w"), indent=1)
print(json.dumps(doc, indent=1))
