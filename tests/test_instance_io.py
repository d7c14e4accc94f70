"""CPU-only: the wasm JSON schema (web/employee-scheduling-wasm-bindgen/src/lib.rs:86-110)."""
import json

import pytest

from ls_amd import instance_io as io

EXAMPLE = json.dumps({
    "startDate": "2022-05-09", "endDate": "2022-06-08",
    "employees": [{"id": 5}, {"id": 0}, {"id": 9}],
    "employeeHolidays": [["2022-05-09", "2022-06-08"], [], ["2022-05-12"]],
})


def test_load_input_maps_ids_and_holidays():
    inp = io.load_input(EXAMPLE)
    assert inp.employee_ids == [0, 5, 9]                      # BTreeSet order
    assert inp.n_days_scored == 31 and inp.start_weekday == 0  # 2022-05-09 is a Monday
    assert int(inp.holiday_bits[1]) == (1 << 0) | (1 << 30)    # id 5 -> index 1
    assert int(inp.holiday_bits[0]) == 0
    assert int(inp.holiday_bits[2]) == 1 << 3


def test_solution_wrapper_format():
    inp = io.load_input(EXAMPLE)
    sol = [i % 3 for i in range(32)]
    w = io.solution_wrapper(inp, (2, 7), sol)
    assert w["score"] == {"hard_score": 2.0, "soft_score": 7.0}
    assert len(w["days_to_employees"]) == 31                  # the stored 32nd entry is not reported
    assert w["days_to_employees"][0] == ["Mon 2022-05-09", {"id": 0}]
    assert w["days_to_employees"][1] == ["Tue 2022-05-10", {"id": 5}]
    assert w["days_to_employees"][30][0] == "Wed 2022-06-08"
    json.loads(io.dumps_solution(inp, (2, 7), sol))


def test_rejects_what_the_reference_panics_on():
    bad = json.loads(EXAMPLE)
    bad["employeeHolidays"][0] = ["2022-07-01"]
    with pytest.raises(ValueError):
        io.load_input(json.dumps(bad))
