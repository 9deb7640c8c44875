"""The SQL texts of the reference's end-to-end tests (/root/reference/tests/test_e2e.py:88-419), keyed by the
name of the catalogue case (tests/queries.py) that restates the same query through the DataFrame API.  The
golden rows of those cases were produced by the real reference, so they pin both routes."""

E2E_SQL = {
    "e2e_select_star": "SELECT * FROM '{users}';",
    "e2e_where_eq_str": "SELECT first_name, last_name FROM '{users}' WHERE country='USA';",
    "e2e_concat": "SELECT first_name + ' ' + last_name AS full_name FROM '{users}';",
    "e2e_int_arith": "SELECT user_id, age, age+5 AS age_in_5_years FROM '{users}';",
    "e2e_where_float_gt": "SELECT * FROM '{orders}' WHERE price > 100;",
    "e2e_int_times_float": "SELECT product, quantity*price AS total_value FROM '{orders}';",
    "e2e_between_ts": "SELECT * FROM '{orders}' WHERE order_date BETWEEN '2025-03-01' AND '2025-06-01';",
    "e2e_like": "SELECT * FROM '{orders}' WHERE product LIKE '%top%';",
    "e2e_group_count": "SELECT country, COUNT() AS user_count FROM '{users}' GROUP BY country;",
    "e2e_group_sum_expr": "SELECT user_id, SUM(quantity*price) AS total_spent FROM '{orders}' GROUP BY user_id;",
    "e2e_group_avg_float": "SELECT product, AVG(price) AS avg_price FROM '{orders}' GROUP BY product;",
    "e2e_group_avg_int": "SELECT country, AVG(age) AS avg_age FROM '{users}' GROUP BY country;",
    "e2e_having_count": "SELECT user_id, COUNT() AS order_count FROM '{orders}' GROUP BY user_id HAVING COUNT() > 1;",
    "e2e_join_select": "SELECT u.first_name, o.product FROM '{users}' AS u JOIN '{orders}' AS o ON u.user_id=o.user_id;",
    "e2e_join_group_count": "SELECT u.country, COUNT() AS orders_count "
                            "FROM '{users}' AS u JOIN '{orders}' AS o ON u.user_id=o.user_id GROUP BY u.country;",
    "e2e_join_group_sum": "SELECT u.first_name, SUM(o.quantity*o.price) AS spent "
                          "FROM '{users}' AS u JOIN '{orders}' AS o ON u.user_id=o.user_id GROUP BY u.first_name;",
    "e2e_join_where_float": "SELECT u.first_name, o.product, o.price "
                            "FROM '{users}' AS u LEFT JOIN '{orders}' AS o ON u.user_id=o.user_id WHERE o.price > 100;",
    "e2e_join_where_ts": "SELECT u.first_name, o.product, o.order_date "
                         "FROM '{orders}' AS o LEFT JOIN '{users}' AS u ON u.user_id=o.user_id "
                         "WHERE o.order_date > '2025-05-01';",
    "e2e_group_sum_max": "SELECT product, SUM(quantity) AS total_quantity, MAX(price) AS max_price FROM '{orders}' "
                         "GROUP BY product;",
    "e2e_join_group_having": "SELECT u.country, COUNT() AS orders_count, SUM(o.quantity*o.price) AS total_sales "
                             "FROM '{users}' AS u JOIN '{orders}' AS o ON u.user_id=o.user_id GROUP BY u.country "
                             "HAVING SUM(o.quantity*o.price) > 500;",
}
